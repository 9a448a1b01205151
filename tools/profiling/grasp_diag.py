"""Per-step trace of the 'grasp' action schedule (bench.py's timed window): contact counters, grasp state, finger forces,
physics time.  usage: grasp_diag.py [config] [envs] [close_at] [steps] [close_rate: 0 = the command jumps]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "real2sim-eval_amd"), R]
import numpy as np
import torch

from r2s_hip.rollout import BatchedRollout

cfg = sys.argv[1] if len(sys.argv) > 1 else "sloth_32env"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
close_at = int(sys.argv[3]) if len(sys.argv) > 3 else 6
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 14
rate = float(sys.argv[5]) if len(sys.argv) > 5 else 0.1
ro = BatchedRollout(cfg, n_env=envs, close_at=close_at, close_rate=rate if rate > 0 else None)
print("N", ro.N, "S", ro.S, "schedule", ro.schedule, "close_at", ro.close_at, "layout", ro.phys.layout_stats())
ro.phys.set_timing(True)
x0 = ro.phys.x.clone()
for t in range(steps):
    ro.step()
    torch.cuda.synchronize()
    ms, k = ro.phys.last_step_ms()
    st = ro.contact_stats()
    f = ro.phys.collision_forces()
    op, gr = ro.phys.eef_state() if not ro.use_pusher else (torch.ones(envs), torch.zeros(envs))
    x = ro.phys.x
    if os.environ.get("R2S_DIAG_CAND") and st["self_collision_candidates"] > 0:
        num, _ = ro.phys.collision_lists()
        nz = num[num > 0].float()
        print(f"   candidates per listed particle: mean {float(nz.mean()):.1f} max {int(nz.max())} p90 {float(nz.quantile(0.9)):.0f}; listed {int((num > 0).sum())}")
    if os.environ.get("R2S_DIAG_DEFER"):
        dc = ro.phys.deferred_counts()
        print(f"   deferred mesh queries per substep: mean {dc[:-1].mean():.1f} max {int(dc[:-1].max())} last {int(dc[-2])}; near flag {int(dc[-1])}")
    if not ro.use_pusher:
        mm = ro.phys.mesh_map
        pads = [float(torch.linalg.norm(f[0][torch.from_numpy(mm == m).to(f.device)][[18, 19, 1]].sum(0))) for m in (0, 1)]
        print(f"   pad forces (faces 18 + 19 + 1) env 0: {pads[0]:9.1f} {pads[1]:9.1f}  grasped per env {gr.tolist()}")
    print(f"step {t:2d}: phys {ms:7.3f} ms ({ms / k * 1e3:6.2f} us/substep) cand {st['self_collision_candidates']:6d} hits {st['mesh_contacts']:5d} "
          f"grasped {st['grasped_envs']} open {float(op[0]):.3f} |F|max {float(f.abs().max()):9.1f} eef_z {float(ro.eef_xyz[0, 2]):.4f} "
          f"top_z {float(x[0, :, 2].max()):.4f} max|dx| {float((x - x0).abs().max()):.4f} finite {bool(torch.isfinite(x).all())} {st['flavour']['kernel']}")
