"""sha256 of the frames a rollout renders (colour + depth) — run under two builds of the library (R2S_HIP_LIB) to show that a kernel
change leaves the images bit-identical.  usage: python tools/probes/image_hash.py [config] [env steps]"""
import sys, os, hashlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [os.path.join(R, 'real2sim-eval_amd'), R]
import torch
from r2s_hip.rollout import BatchedRollout
cfg = sys.argv[1] if len(sys.argv) > 1 else "sloth_32env"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ro = BatchedRollout(cfg, num_substeps=40)
h = hashlib.sha256()
for _ in range(n):
    ro.step()
    col, dep = ro.observations()
    torch.cuda.synchronize()
    h.update(col.cpu().numpy().tobytes()); h.update(dep.cpu().numpy().tobytes())
print(cfg, "frames", tuple(col.shape), "instances", ro.last_num_rendered, "sha256", h.hexdigest())
