"""Round 5: the resident stepper's self-collision flavour on a grasp of the toy by ONE environment (finger meshes in reach AND the arms
pressed together: candidates + mesh contact in the same launch), against the per-substep flavour of the same handle (R2S_RES_SELF=0):
flavours taken, agreement, wall time per env step.  `python tools/probes/resident_self_probe.py [config] [n_env] [steps]`."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "real2sim-eval_amd"))
import numpy as np, torch


def run(cfg, n_env, steps, res_self):
    os.environ["R2S_RES_SELF"] = "1" if res_self else "0"
    from r2s_hip.rollout import BatchedRollout
    ro = BatchedRollout(cfg, n_env=n_env, close_at=2)
    xs, fl, ms = [], [], []
    for _ in range(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ro.physics_step(); ro.t += 1
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
        xs.append(ro.phys.x.cpu().numpy().copy()); fl.append(ro.phys.last_flavour()["kernel"])
    st = ro.contact_stats()
    ro.phys.step()
    torch.cuda.synchronize()
    return xs, fl, ms, st


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "sloth_32env"
    n_env = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    a = run(cfg, n_env, steps, True)
    b = run(cfg, n_env, steps, False)
    c = run(cfg, n_env, steps, True)
    for k in range(steps):
        print(f"step {k}: {a[2][k]:7.2f} ms  {a[1][k]:60s} | {b[2][k]:7.2f} ms  {b[1][k]:60s} | dx {np.abs(a[0][k] - b[0][k]).max():.2e} | rerun equal {np.array_equal(a[0][k], c[0][k])}")
    print("stats", a[3], b[3])
