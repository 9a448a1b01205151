import sys, os
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'real2sim-eval_amd'),R]
import torch, numpy as np
from r2s_hip.rollout import BatchedRollout
from r2s_hip.raster import _memcpy_d2d
ro = BatchedRollout("sloth_32env", num_substeps=2, self_collision=False)
ro.step(); torch.cuda.synchronize()
T = torch.empty(64, 480, 640, device="cuda"); N = torch.empty(64, 480, 640, dtype=torch.int32, device="cuda")
ro.raster.set_aux(T, N)
ro.render(); torch.cuda.synchronize()
dbg = ro.raster.debug(); tiles = 40*30
r = torch.empty(64*tiles, 2, dtype=torch.int32, device="cuda"); _memcpy_d2d(r.data_ptr(), dbg["ranges_ptr"], r.numel()*4, "cuda:0")
lens = (r[:,1]-r[:,0]).cpu().numpy().reshape(64, 30, 40)
print("percentiles 50/90/99/99.9/max", np.percentile(lens, [50,90,99,99.9,100]))
for v in (0,1): 
    l = lens[v::2]; print("view", v, "mean", l.mean().round(1), "max", l.max(), "sum", l.sum())
# last contributor per pixel -> per-tile max = how deep the tile's list was actually walked
n = N.cpu().numpy().reshape(64, 30, 16, 40, 16).max(axis=(2,4))
print("walked depth: mean", n.mean().round(1), "max", n.max(), " ratio walked/len (tiles with len>2000):", (n[lens>2000]/lens[lens>2000]).mean().round(3), "count", (lens>2000).sum())
big = np.argwhere(lens == lens.max())[0]; print("largest tile at frame,ty,tx", big, "walked", n[tuple(big)])
print("sum of walked depth", n.sum(), "vs sum of len", lens.sum())
