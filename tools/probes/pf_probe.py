"""In-kernel wall-clock picture of ONE k_substep_pf launch (the last substep of an env step in contact; -DR2S_PHASE_PROBE build via
R2S_HIP_LIB): when the finishers at its head delivered, when the fused blocks entered, which of them waited for a finisher and for
how long, when the launch ended.  usage: pf_probe.py [config] [envs] [close_at] [steps] [close_rate]   (R2S_CHAINS=1 for a collision-free table;
close_rate: the closing ramp of round 6 — with it the grasp latches and the probed launch is one of the HOLD state)"""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "real2sim-eval_amd"), R]
import numpy as np
import torch

from r2s_hip import _lib
from r2s_hip.rollout import BatchedRollout

cfg = sys.argv[1] if len(sys.argv) > 1 else "sloth_32env"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
close_at = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
rate = float(sys.argv[5]) if len(sys.argv) > 5 else None
ro = BatchedRollout(cfg, n_env=envs, close_at=close_at, close_rate=rate)
for _ in range(steps):
    ro.physics_step(); ro.t += 1
torch.cuda.synchronize()
L = _lib.lib()
fl = ro.phys.last_flavour()
print("flavour", fl["kernel"], "chains", fl["chains"], "deferred (last-but-one substep)", int(ro.phys.deferred_counts()[-3]), ro.contact_stats()["self_collision_candidates"], "candidates")
nq = 1024
qb = (C.c_longlong * (nq * 32))()
L.r2s_phys_debug_query_probe.argtypes = [C.c_void_p, C.c_int]
L.r2s_phys_debug_query_probe(qb, nq)
q = np.array(qb, dtype=np.int64).reshape(nq, 32).astype(np.float64) * 0.01
nb = 8192
pb = (C.c_longlong * (nb * 4))()
L.r2s_phys_debug_phase_probe.argtypes = [C.c_void_p, C.c_int]
L.r2s_phys_debug_phase_probe(pb, nb)
a = np.array(pb, dtype=np.int64).reshape(nb, 4).astype(np.float64) * 0.01
blocks = a[(a[:, 0] > 0) & (a[:, 3] > 0)]
# the tables keep rows of earlier launches where the last one wrote none (other grid sizes, the settling steps): only the LAST launch counts
t_end = blocks[:, 3].max()
blocks = blocks[blocks[:, 0] > t_end - 100.0]
q[(q[:, 31] > 0) & (q[:, 31] < t_end - 100.0)] = 0
q[512:][(q[512:, 30] > 0) & (q[512:, 30] < t_end - 100.0)] = 0
# the finishers' stamps are of substep n_sub - 2 = the head of the LAST launch when the flavour is k_substep_pf (else: of their own launch)
part1 = q[:512][q[:512, 0] > 0]
last = np.array([r[:28][r[:28] > 1000].max() for r in part1]) if len(part1) else np.array([])
entry1 = part1[:, 31] if len(part1) else np.array([])
p2 = q[512:][(q[512:, 28] > 0) & (q[512:, 27] > 0)]
t0 = min([blocks[:, 0].min()] + ([entry1.min()] if len(entry1) else []) + ([p2[:, 30].min()] if len(p2) else []))
pc = lambda v, ps=(0, 50, 90, 100): np.round(np.percentile(v, ps), 2)  # noqa: E731
print(f"times: us since the launch's first wavefront entered.  fused blocks with stamps {len(blocks)}; launch span (first entry -> last end) {blocks[:, 3].max() - t0:.2f} us")
print("percentiles shown: min / median / p90 / max")
if len(part1):
    print("finishers, part 1 (mesh particles): wavefronts", len(part1), "entered at", pc(entry1 - t0), "delivered at", pc(last - t0))
if len(part1):  # stamp by stamp (batched small-scene finishers: staged, first query back, second query back, stored[, the next batch ...])
    # batched finisher (round 6), per wavefront: 0 staged, 1 record + impulses | first query: 2 bounds, 3 upper bound, 4 packed, 5 = surviving pairs (COUNT), 6 evaluated,
    # 7 = lanes that need the generic winding number (COUNT) | 8 query back | re-query: 9, 10 (skipped steps), 11 packed, 12 pairs (COUNT), 13 evaluated, 14 generic lanes (COUNT) |
    # 15 query back | 16 stored
    names = {0: "staged", 1: "record + impulses", 2: "q1 bounds", 3: "q1 upper bound", 4: "q1 packed", 5: "q1 surviving pairs (COUNT)", 6: "q1 evaluated", 7: "q1 lanes on the generic winding number (COUNT)",
             8: "q1 back", 9: "q2 -", 10: "q2 -", 11: "q2 packed", 12: "q2 surviving pairs (COUNT)", 13: "q2 evaluated", 14: "q2 lanes on the generic winding number (COUNT)", 15: "q2 back", 16: "stored"}
    counts = (5, 7, 12, 14)
    batched = len(part1) and (part1[:, 5] * 100 < 4096).all()
    for k in range(17 if batched else 8):
        col = part1[:, k]
        if (col > 0).sum() == 0 and not (batched and k in counts):
            break
        if batched and k in counts:
            print(f"  part 1 stamp {k} {names[k]}:", np.round(np.percentile(col * 100, (0, 50, 90, 100)), 0))
        else:
            print(f"  part 1 stamp {k} {names.get(k, '') if batched else ''}: wavefronts {int((col > 0).sum())} at", pc(col[col > 0] - t0))
if len(p2):
    print("finishers, part 2 (candidates): busy wavefronts", len(p2), "entered the kernel at", pc(p2[:, 30] - t0), "left part 2 at", pc(p2[:, 29] - t0))
b = blocks - t0
print("fused blocks: entered at", pc(b[:, 0]), "staged at", pc(b[:, 1]), "ended at", pc(b[:, 3]))
stage = b[:, 1] - b[:, 0]
w = stage > np.median(stage) + 2.0
print(f"blocks whose staging took > median + 2 us (waited for a finisher): {int(w.sum())} of {len(b)}; median staging of the others {np.median(stage[~w]):.2f} us")
if w.any():
    print("  waiting blocks: entered at", pc(b[w, 0]), "staged at", pc(b[w, 1]), "ended at", pc(b[w, 3]), "gather + finish after staging", pc(b[w, 3] - b[w, 1]))
    print("  the others: ended at", pc(b[~w, 3]), "gather + finish after staging", pc(b[~w, 3] - b[~w, 1]))
    print("  waiting blocks: gather", pc(b[w, 2] - b[w, 1]), "finish (tests, pushes, ground, store)", pc(b[w, 3] - b[w, 2]), "| the others: gather", pc(b[~w, 2] - b[~w, 1]), "finish", pc(b[~w, 3] - b[~w, 2]))
