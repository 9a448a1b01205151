"""Where a resident launch with owning query servers spends its time while the gripper holds the rope: wall clock by phase of wavefront 0
of every block (as resident_probe.py) and of the first wavefront of every busy server pair, summed over one launch (667 substeps).
Needs a library built with -DR2S_PHASE_PROBE (tools/profiling/build_variants.sh probe "-DR2S_PHASE_PROBE", R2S_HIP_LIB=scratch/variants/libr2s_probe.so)."""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [os.path.join(R, 'real2sim-eval_amd'), R]
import torch, numpy as np
from r2s_hip.rollout import BatchedRollout
from r2s_hip import _lib
ro = BatchedRollout("rope_1env", close_at=4, seed=0)
ro.phys.set_timing(True)
for _ in range(int(os.environ.get("STEPS", "16"))):
    ro.step()
torch.cuda.synchronize()
ms, k = ro.phys.last_step_ms()
print("flavour", ro.phys.last_flavour()["kernel"], "us/substep %.2f" % (ms / k * 1e3))
L = _lib.lib()
L.r2s_phys_debug_phase_probe.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_longlong * (8192 * 4))()
print("rc", L.r2s_phys_debug_phase_probe(buf, 8192))
a = np.array(buf, dtype=np.int64).astype(np.float64)
n = ro.phys.layout_stats()["blocks"] * ro.n_env
blk = a[:n * 8].reshape(n, 8)
us = blk[:, :4] * 0.01 / 667
print("blocks (wavefront 0), us per substep: poll %.2f  gather+reduce %.2f  finish(+result wait) %.2f  publish %.2f; poll passes %.2f" % (*us.mean(0), blk[:, 4].mean() / 666))
busy = np.argsort(-us[:, 2])[:6]
print("  blocks with the longest finish phase:", [(int(b), [round(float(x), 2) for x in us[b]]) for b in busy])
srv = a[16384:16384 + 1024 * 8].reshape(1024, 8)
srv = srv[srv[:, 4] > 0]
if len(srv):
    per = srv[:, :4] * 0.01 / srv[:, 4:5]
    print("server pairs busy: %d; substeps served per pair: mean %.0f" % (len(srv), srv[:, 4].mean()))
    print("  us per served substep: wait %.2f  force+sum %.2f  finish %.2f  store %.2f  (sum %.2f); poll passes %.2f" % (*per.mean(0), per.sum(1).mean(), (srv[:, 5] / srv[:, 4]).mean()))
    print("  max over pairs:", [round(float(x), 2) for x in per.max(0)], " min:", [round(float(x), 2) for x in per.min(0)])
    L.r2s_phys_debug_query_probe.argtypes = [C.c_void_p, C.c_int]
    qb = (C.c_longlong * (1024 * 32))()
    L.r2s_phys_debug_query_probe(qb, 1024)
    q = np.array(qb, dtype=np.int64).reshape(1024, 32)[:len(srv), :4].astype(np.float64)
    d = np.diff(q, axis=1) * 0.01
    d = d[(d > 0).all(1) & (d < 50).all(1)]
    print("  last substep of a pair, us: first query %.2f  response + second query %.2f  rest of the substep %.2f  (%d pairs)" % (*d.mean(0), len(d)))
