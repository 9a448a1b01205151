"""Soak run: N env steps of a config; checks the state stays finite, reports env-steps/s per 50-step window and GPU memory."""
import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'real2sim-eval_amd'),R]
import torch
from r2s_hip.rollout import BatchedRollout
cfg = sys.argv[1] if len(sys.argv) > 1 else "sloth_32env"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ro = BatchedRollout(cfg)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(n):
    ro.step()
    if k % 50 == 49:
        torch.cuda.synchronize(); t1 = time.perf_counter()
        x = ro.phys.x
        print(f"step {k+1}: {ro.n_env*50/(t1-t0):8.1f} env-steps/s | finite {bool(torch.isfinite(x).all())} | z min {float(x[...,2].min()):+.4f} max |v| {float(ro.phys.v.abs().max()):.3f} "
              f"| instances {ro.last_num_rendered} | mem {torch.cuda.memory_allocated()/2**20:.0f} MiB torch, {torch.cuda.mem_get_info()[0]/2**30:.1f} GiB free | success flags {int(ro.success_flags().sum())} | lossy raster batches {ro.lossy_batches}")
        t0 = time.perf_counter()
