import sys, time, os
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'real2sim-eval_amd'),R]
import torch, numpy as np
from r2s_hip.rollout import BatchedRollout
cfg = sys.argv[1] if len(sys.argv)>1 else "sloth_32env"
nsub = int(sys.argv[2]) if len(sys.argv)>2 else 667
reps = int(sys.argv[3]) if len(sys.argv)>3 else 3
ro = BatchedRollout(cfg, num_substeps=nsub)
print("layout", ro.phys.layout_stats(), "N", ro.N, "S", ro.S)
ro.phys.set_timing(True)
for r in range(reps):
    ro.physics_step(); torch.cuda.synchronize()
    ms,k = ro.phys.last_step_ms(); print(f"rep {r}: {ms:.3f} ms / {k} kernels = {ms/k*1e3:.2f} us")
