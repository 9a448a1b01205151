"""Soak of the hand-offs inside k_substep_pf (finishers at the head of the next launch): the same rollout through a grasp N times — the
headline batch (32 environments, four chains) or the pusher's — each time next to a second stream that keeps the chip streaming through
HBM; every run must end in the same bits as the first and none may run a poll into its limit (a sticky fault raises at the next step).
`rope_fold_1env` puts the same soak on the resident stepper's self-collision flavour (two tagged hand-offs per substep, 3 000 particles
with candidates).
usage: soak_pf.py [config] [runs] [steps] [close_rate]   (round 6: with a close_rate the grasp latches — the batched finishers and the held grasp are in the soak)"""
import hashlib
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "real2sim-eval_amd"), R]
import torch

from r2s_hip.rollout import BatchedRollout

cfg = sys.argv[1] if len(sys.argv) > 1 else "sloth_32env"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
side = torch.cuda.Stream()
big = torch.empty(1 << 27, dtype=torch.float32, device="cuda")
rate = float(sys.argv[4]) if len(sys.argv) > 4 else None
ro = BatchedRollout(cfg, close_at=2, close_rate=rate)
x0, v0 = ro._init["x"].clone(), ro._init["v"].clone()
ref, bad, substeps = None, 0, 0
for r in range(runs):
    ro.t = 0
    ro.reset()                                    # every environment back to the start state (a full reset: new flavour history); the trace restarts at t = 0
    for k in range(steps):
        with torch.cuda.stream(side):
            for _ in range(3 + (r + k) % 4):      # uneven, different every step
                big.mul_(1.0001)
        ro.physics_step(); ro.t += 1
    torch.cuda.synchronize()
    ro.phys.step(0, 0); ro.t += 1                 # raises on a sticky fault
    torch.cuda.synchronize()
    h = hashlib.sha1(ro.phys.x.cpu().numpy().tobytes() + ro.phys.v.cpu().numpy().tobytes()).hexdigest()[:16]
    substeps += (steps + 1) * 667
    if ref is None:
        ref = h
        st = ro.contact_stats()
        print("flavour", ro.phys.last_flavour()["kernel"], {k: st[k] for k in ("self_collision_candidates", "mesh_contacts", "grasped_envs")})
        assert st["mesh_contacts"] > 0 or st["self_collision_candidates"] > 0, "the soak must run in contact"
    bad += h != ref
    if h != ref:
        print("run", r, "differs:", h, "vs", ref)
print(f"{cfg}: {runs} runs x {steps + 1} env steps = {substeps} launches per chain, {bad} runs differ from the first, state hash {ref}, no fault")
