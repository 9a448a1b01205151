import sys, os
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'real2sim-eval_amd'),R]
import torch
from r2s_hip.rollout import BatchedRollout
ro = BatchedRollout("sloth_32env", num_substeps=20)
for _ in range(3): ro.step()
torch.cuda.synchronize()
