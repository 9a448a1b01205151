import sys, os, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'real2sim-eval_amd'),R]
import torch
from r2s_hip.rollout import BatchedRollout
from r2s_hip import _lib
ro = BatchedRollout("sloth_32env", num_substeps=2, self_collision=False)
ro.step()
L=_lib.lib(); buf=(C.c_ulonglong*4)()
L.r2s_raster_debug_comp_stats(buf, 1)
ro.render(); torch.cuda.synchronize()
L.r2s_raster_debug_comp_stats(buf, 1)
it, hits, zero, alive = [int(v) for v in buf]
Lr = ro.last_num_rendered
print(f"instances {Lr}, wave-iterations {it} ({it/Lr:.2f} per instance of max 4), zero-hit iterations {zero/it:.1%}, hit lanes / (64*iterations) {hits/(64*it):.1%}, alive lanes {alive/(64*it):.1%}")
