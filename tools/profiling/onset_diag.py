"""Per-step wall time and flavour of ONE environment of a config through a contact onset (small batches: resident launch with query servers ->
per-substep kernels when the units run low).  usage: onset_diag.py [config] [steps] [close_at] [close_rate]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "real2sim-eval_amd"), R]
import torch

from r2s_hip.rollout import BatchedRollout

cfg = sys.argv[1] if len(sys.argv) > 1 else "sloth_32env"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 18
close_at = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rate = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
ro = BatchedRollout(cfg, n_env=1, close_at=close_at, close_rate=rate, seed=1)
ro.phys.set_timing(True)
for t in range(steps):
    ro.physics_step()
    torch.cuda.synchronize()
    ms = ro.phys.last_step_ms()[0]
    st = ro.contact_stats()
    dc = ro.phys.deferred_counts()
    print(f"step {t:2d}: {ms:8.2f} ms  cand {st['self_collision_candidates']:5d} hits {st['mesh_contacts']:4d} grasped {st['grasped_envs']} deferred/substep max {int(dc[:-1].max()):4d}  {ro.phys.last_flavour()['kernel'][:110]}")
    ro.t += 1
