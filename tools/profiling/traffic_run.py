import sys, os
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'real2sim-eval_amd'),R]
import torch
from r2s_hip.rollout import BatchedRollout
# calibration kernels with known HBM byte counts (working set >> 256 MiB Infinity Cache)
x = torch.empty(512*1024*1024//4, device="cuda").normal_(); y = torch.empty_like(x)   # 512 MiB each
for _ in range(3): y.copy_(x)          # reads 512 MiB, writes 512 MiB per call
torch.cuda.synchronize()
ro = BatchedRollout("sloth_32env", num_substeps=40)
for _ in range(2): ro.step()
torch.cuda.synchronize()
