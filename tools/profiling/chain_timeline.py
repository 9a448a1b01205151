#!/usr/bin/env python
"""Per-chain timeline of the fused substep kernels from a rocprofv3 rocpd database (kernel-trace): for every hardware queue, the
kernel duration, the start-to-start interval and the gap between the end of one substep's kernel and the start of the next one's on the
SAME queue, plus how many substep kernels run at once on average.  Says whether a batched substep is paced by the kernels or by what
happens between them.  Usage: python tools/profiling/chain_timeline.py <results.db> [name filter, default k_substep]"""
import sqlite3, sys
import numpy as np

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "k_substep"
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
print("columns:", cols)
rows = cur.execute(f"select {name}, start, end, {qcol if qcol else 0} from kernels order by start").fetchall()
rows = [r for r in rows if pat in r[0]]
byq = {}
for n, s, e, q in rows:
    byq.setdefault(q, []).append((s, e, n))
print(f"{len(rows)} kernels matching '{pat}' on {len(byq)} queues ({qcol})")
pct = lambda a: "p10 %.2f / p50 %.2f / p90 %.2f" % tuple(np.percentile(a, [10, 50, 90]))
for q, ks in sorted(byq.items()):
    ks.sort()
    s = np.array([k[0] for k in ks], float); e = np.array([k[1] for k in ks], float)
    dur = (e - s) * 1e-3; gap = (s[1:] - e[:-1]) * 1e-3; itv = (s[1:] - s[:-1]) * 1e-3
    ok = gap < 200   # inside an env step
    print(f"queue {q}: {len(ks)} kernels; duration us {pct(dur)}; gap to the next on this queue us {pct(gap[ok])}; start-to-start us {pct(itv[ok])}")
# concurrency: time-weighted number of matching kernels in flight while at least one is
ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
t_prev, level, acc, busy = ev[0][0], 0, 0.0, 0.0
hist = {}
for t, d in ev:
    if level > 0 and t - t_prev < 200e3:
        acc += level * (t - t_prev); busy += t - t_prev
        hist[level] = hist.get(level, 0) + (t - t_prev)
    level += d; t_prev = t
print("kernels in flight while any is: mean %.2f; share of time by level:" % (acc / busy), {k: round(v / busy, 3) for k, v in sorted(hist.items())})
