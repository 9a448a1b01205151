"""Raster stage times of the bench's render for a config (min of a few timed renders)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [os.path.join(R, 'real2sim-eval_amd'), R]
import torch
from r2s_hip.rollout import BatchedRollout
cfg = sys.argv[1] if len(sys.argv) > 1 else "sloth_32env"
ro = BatchedRollout(cfg, num_substeps=20)
for _ in range(3):
    ro.step()
torch.cuda.synchronize()
ro.raster.set_timing(True)
best = None
for _ in range(5):
    ro.render(); torch.cuda.synchronize()
    st = ro.raster.stage_ms()
    best = st if best is None else {k: min(best[k], v) for k, v in st.items()}
print(cfg, {k: round(v, 3) for k, v in best.items()}, "sum %.3f" % sum(best.values()))
