"""Tile list lengths of one rendered batch, per frame (why the compositor of a single environment takes as long as it does)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [os.path.join(R, 'real2sim-eval_amd'), R]
import torch, numpy as np
from r2s_hip.rollout import BatchedRollout
from r2s_hip.raster import _memcpy_d2d
cfg = os.environ.get("VB_CONFIG", "rope_1env")
ro = BatchedRollout(cfg, num_substeps=20)
for _ in range(3):
    ro.step()
ro.raster.set_timing(True)
ro.render(); torch.cuda.synchronize()
print("stage ms", ro.raster.stage_ms())
dbg = ro.raster.debug()
tiles = ((ro.W + 15) // 16) * ((ro.H + 15) // 16)
frames = ro.n_env * ro.views
rng_t = torch.empty(frames * tiles, 2, dtype=torch.int32, device=ro.device)
_memcpy_d2d(rng_t.data_ptr(), dbg["ranges_ptr"], rng_t.numel() * 4, ro.device)
lens = (rng_t[:, 1] - rng_t[:, 0]).cpu().numpy().reshape(frames, tiles)
for f in range(min(frames, 4)):
    l = np.sort(lens[f])[::-1]
    print(f"frame {f}: instances {l.sum()}, top tiles {l[:12].tolist()}, tiles > 256: {(l > 256).sum()}, > 1024: {(l > 1024).sum()}, rounds total {np.ceil(l / 256).sum():.0f}")
