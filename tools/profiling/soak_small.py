"""Soak run of a one-environment rollout (the resident stepper and its flavour switches): N env steps of a config with n_env = 1; per 50
steps: env-steps/s, share of steps that ran as one resident launch, finiteness; at the end the same rollout with R2S_RESIDENT=0 semantics
(set_resident(False) from the start) for comparison of the final state."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [os.path.join(R, 'real2sim-eval_amd'), R]
import torch, numpy as np
from r2s_hip.rollout import BatchedRollout
cfg = sys.argv[1] if len(sys.argv) > 1 else "rope_1env"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300

def run(resident):
    ro = BatchedRollout(cfg, n_env=1, close_at=int(os.environ.get("CLOSE_AT", "40")))
    if not resident:
        ro.phys.set_resident(False)
    res = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        ro.step()
        res += bool(ro.phys.last_flavour().get("resident"))
        if k % 50 == 49:
            torch.cuda.synchronize(); t1 = time.perf_counter()
            x = ro.phys.x
            st = ro.contact_stats()
            print(f"  step {k+1}: {50/(t1-t0):7.1f} env-steps/s | resident steps so far {res} | finite {bool(torch.isfinite(x).all())} | max |v| {float(ro.phys.v.abs().max()):.3f} "
                  f"| mesh contacts {st['mesh_contacts']} candidates {st['self_collision_candidates']} | flavour {st['flavour']['kernel']}", flush=True)
            t0 = time.perf_counter()
    torch.cuda.synchronize()
    return ro.phys.x.cpu().numpy().copy()

print("resident stepper on:"); xa = run(True)
print("resident stepper off:"); xb = run(False)
print("final state, max |dx| between the two runs: %.3e m (chaotic amplification of last-bit differences over %d x 667 substeps included)" % (np.abs(xa - xb).max(), n))
