"""In-kernel wall-clock stamps of k_contact_finish's wavefronts (one per particle in contact), built with -DR2S_PHASE_PROBE
(scratch/libr2s_probe.so via R2S_HIP_LIB).  usage: query_probe.py [config] [envs] [close_at] [steps]"""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "real2sim-eval_amd"), R]
import numpy as np
import torch

from r2s_hip import _lib
from r2s_hip.rollout import BatchedRollout

cfg = sys.argv[1] if len(sys.argv) > 1 else "T_pusher_32env"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
close_at = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ro = BatchedRollout(cfg, n_env=envs, close_at=close_at)
if os.environ.get("R2S_CHAINS"):
    ro.phys.set_tuning(int(os.environ["R2S_CHAINS"]), -1)
for _ in range(steps):
    ro.step()
torch.cuda.synchronize()
L = _lib.lib()
n = 1024
buf = (C.c_longlong * (n * 32))()
L.r2s_phys_debug_query_probe.argtypes = [C.c_void_p, C.c_int]
print("rc", L.r2s_phys_debug_query_probe(buf, n), "flavour", ro.phys.last_flavour()["kernel"], "deferred (last-but-one substep)", int(ro.phys.deferred_counts()[-3]))
a = np.array(buf, dtype=np.int64).reshape(n, 32).astype(np.float64) * 0.01  # us (100 MHz)
p2 = a[512:][a[512:, 28] > 0]   # rows 512..: part 2's wavefronts counted from the END of the grid (the busy ones)
if len(p2):
    busy = p2[p2[:, 27] > 0]
    print("part 2 (candidate particles, 16 lanes each): wavefronts recorded", len(p2), "busy", len(busy))
    if len(busy):
        cnt = busy[:, 27] * 100.0   # (raw integer: undo the 0.01 scaling)
        dur = busy[:, 29] - busy[:, 28]
        print("  busy wavefronts: entered part 2 at (us since their kernel entry) median/p90/max", np.round(np.percentile(busy[:, 28] - busy[:, 30], [50, 90, 100]), 2),
              "left at", np.round(np.percentile(busy[:, 29] - busy[:, 30], [50, 90, 100]), 2))
        print("  largest candidate count per wavefront: median/p90/max", np.percentile(cnt, [50, 90, 100]), " time in part 2 (us) median/p90/max", np.round(np.percentile(dur, [50, 90, 100]), 2))
        for lo, hi in ((1, 16), (17, 32), (33, 64), (65, 128), (129, 500)):
            m = (cnt >= lo) & (cnt <= hi)
            if m.any():
                print(f"    count {lo}-{hi}: {int(m.sum())} wavefronts, part-2 time median {np.median(dur[m]):.2f} us")
used = (a[:512, 0] > 0)
a = a[:512][used]
print("wavefronts with stamps:", len(a))
if len(a):
    entry = a[:, 31]
    t0 = entry.min()
    nst = int((a[:, :28] > 0).sum(1).max())
    rel = a[:, :nst] - entry[:, None]
    rel[a[:, :nst] <= 0] = np.nan
    print("kernel entry spread (us):", np.round(np.percentile(entry - t0, [0, 50, 100]), 2))
    names = "stamps in program order: loaded | [clusters | tri round ... | sign] first query | first query back | [second query ...] | response + second query back | stored"
    print(names)
    print("median us since entry:", np.round(np.nanmedian(rel, 0), 2))
    print("p90    us since entry:", np.round(np.nanpercentile(rel, 90, 0), 2))
    print("stamps per wavefront: min", int((a[:, :28] > 0).sum(1).min()), "max", nst)
    print("first wavefront:", np.round(rel[0], 2))
