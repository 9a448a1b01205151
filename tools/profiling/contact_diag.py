"""Per-env-step diagnosis of the contact path on the headline scene: graph time per substep, particles handed to the finishing
kernel per substep (max / mean over the step), tagged entries, candidates, mesh hits — through the closing step and the hold."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [os.path.join(R, "real2sim-eval_amd"), R]
import numpy as np, torch
from r2s_hip.rollout import BatchedRollout
n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dephase = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ro = BatchedRollout("sloth_32env", n_env=n_env, num_substeps=667, close_at=3)
if dephase > 1:
    ro.set_dephase(dephase)
ro.phys.set_timing(True)
for t in range(12):
    ro.physics_step(); ro.t += 1
    torch.cuda.synchronize()
    ms, k = ro.phys.last_step_ms()
    dc = ro.phys.deferred_counts()[:-1]
    st = ro.contact_stats()
    print(f"step {t:2d}: {ms / k * 1e3:6.2f} us/substep  listed/substep max {int(dc.max()):5d} mean {dc.mean():7.1f}  tagged {ro.phys.tagged_count():5d}  "
          f"cand {st['self_collision_candidates']:5d} hits {st['mesh_contacts']:4d}  {st['flavour']['kernel']}", flush=True)
