"""Where a resident launch (k_steps_resident) spends its time: per-workgroup wall clock of wavefront 0 by phase, summed over the
667 substeps.  Needs a library built with -DR2S_PHASE_PROBE (tools/profiling/build_variants.sh probe "-DR2S_PHASE_PROBE",
R2S_HIP_LIB=scratch/variants/libr2s_probe.so)."""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [os.path.join(R, 'real2sim-eval_amd'), R]
import torch, numpy as np
from r2s_hip.rollout import BatchedRollout
from r2s_hip import _lib
cfg = os.environ.get("VB_CONFIG", "rope_1env")
ro = BatchedRollout(cfg, num_substeps=667, n_env=int(os.environ.get("N_ENV", "1")))
ro.phys.set_timing(True)
for _ in range(int(os.environ.get("STEPS", "3"))):
    ro.physics_step(); ro.t += 1
torch.cuda.synchronize()
ms, k = ro.phys.last_step_ms()
print("flavour", ro.phys.last_flavour()["kernel"], "us/substep", ms / k * 1e3)
L = _lib.lib()
n = ro.phys.layout_stats()["blocks"] * ro.n_env
buf = (C.c_longlong * (n * 8))()
L.r2s_phys_debug_phase_probe.argtypes = [C.c_void_p, C.c_int]
print("rc", L.r2s_phys_debug_phase_probe(buf, n * 2))
a = np.array(buf, dtype=np.int64).reshape(n, 8).astype(np.float64)
us = a[:, :4] * 0.01 / 667
print("per substep, us: poll %.2f  gather+reduce %.2f  finish %.2f  publish %.2f   (mean over %d workgroups); poll passes per substep %.2f" % (*us.mean(0), n, a[:, 4].mean() / 666))
print("max over workgroups:", us.max(0), "min:", us.min(0))
print("shader clock over the launch: %.0f MHz (cycle counter / 100 MHz wall clock)" % (a[:, 5].sum() / a[:, 6].sum() * 100))
act = us[:, 0] > 0
if act.any() and (~act).any():
    print("workgroups with a stamp in phase 0: %d of %d; their medians: %s" % (act.sum(), n, np.median(us[act], 0)))
