import sys, os
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'real2sim-eval_amd'),R]
import torch
from r2s_hip.rollout import BatchedRollout
cfg, n_env = sys.argv[1], int(sys.argv[2])
ro = BatchedRollout(cfg, n_env=n_env, views=1)
ro.phys.set_timing(True)
for r in range(5):
    ro.physics_step(); torch.cuda.synchronize()
    ms, k = ro.phys.last_step_ms()
print(f"{cfg} envs={n_env} chains={ro.phys.layout_stats()['chains']}: {ms:.3f} ms = {ms/k*1e3:.2f} us/substep")
