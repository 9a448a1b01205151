"""Host-side cost of one closed-loop env step (what the GPU waits for between the observation and the next step's first kernel):
wall time of the enqueue calls with the device drained before each step."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [os.path.join(R, 'real2sim-eval_amd'), R]
import torch, numpy as np
from r2s_hip.rollout import BatchedRollout
cfg = os.environ.get("VB_CONFIG", "sloth_32env")
ro = BatchedRollout(cfg, num_substeps=667)
for _ in range(4):
    ro.step(); ro.observations()
acc = {}
def tick(name, t0):
    acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
for _ in range(8):
    torch.cuda.synchronize()
    t = time.perf_counter(); act = ro.synthetic_action(ro.t); tick("synthetic_action", t)
    t = time.perf_counter(); ro.apply_action(act); tick("apply_action (enqueue)", t)
    t = time.perf_counter(); ro.phys.step(0, 0, sync_state=True); tick("phys.step (graph launch + state copy enqueue)", t)
    t = time.perf_counter(); torch.cuda.synchronize(); tick("... device time of the step", t)
    t = time.perf_counter(); ro.phys.update_collision_graph(); tick("update_collision_graph (enqueue)", t)
    t = time.perf_counter(); ro.render(); tick("render (enqueue)", t)
    t = time.perf_counter(); torch.cuda.synchronize(); tick("... device time of rebuild + render", t)
    ro.t += 1
for k, v in acc.items():
    print(f"{k:50s} median {np.median(v):7.3f} ms   min {min(v):7.3f}")
