import sys, os, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'real2sim-eval_amd'),R]
import torch, numpy as np
from r2s_hip.rollout import BatchedRollout
from r2s_hip import _lib
ro = BatchedRollout("sloth_32env", num_substeps=667)
ro.physics_step(); ro.physics_step(); torch.cuda.synchronize()
L = _lib.lib()
n = int(os.environ.get("NB", "1888"))
buf = (C.c_longlong * (n * 4))()
L.r2s_phys_debug_phase_probe.argtypes = [C.c_void_p, C.c_int]
print("rc", L.r2s_phys_debug_phase_probe(buf, n))
a = np.array(buf, dtype=np.int64).reshape(n, 4).astype(np.float64) * 0.01  # us (100 MHz)
t0 = a[:, 0].min()
a -= t0
print("kernel span us:", a[:, 3].max())
print("entry   p0/50/100:", np.percentile(a[:, 0], [0, 50, 90, 100]))
print("staged  p0/50/100:", np.percentile(a[:, 1], [0, 50, 90, 100]))
print("springs p0/50/100:", np.percentile(a[:, 2], [0, 50, 90, 100]))
print("end     p0/50/100:", np.percentile(a[:, 3], [0, 50, 90, 100]))
d = np.diff(a, axis=1)
print("phase durations (stage, springs, finish) mean:", d.mean(0), "p90:", np.percentile(d, 90, axis=0), "max:", d.max(0))
late = a[:, 0] > 5
print("blocks entering after 5us:", int(late.sum()), "their mean durations:", d[late].mean(0) if late.any() else None)
np.save(os.path.join(R, "gpurun_out", "phase_%s.npy" % os.environ.get("TAG", "x")), a)
print("layout", ro.phys.layout_stats())
