import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'real2sim-eval_amd'),R]
import torch
from r2s_hip.rollout import BatchedRollout
ro = BatchedRollout(sys.argv[1] if len(sys.argv) > 1 else "sloth_32env", num_substeps=20)
ro.physics_step(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ro.phys.update_collision_graph()
e1.record(); torch.cuda.synchronize()
print("update_collision_graph ms:", e0.elapsed_time(e1) / 10)
