#!/usr/bin/env python
"""Timeline of one env step from a rocprofv3 rocpd database (kernel-trace): what runs between two consecutive physics steps, in time
order, with the idle gaps of the GPU (no kernel running) — where a closed-loop env step spends the time that is neither substeps nor
raster stages.  Usage: python tools/profiling/rocpd_timeline.py <results.db> [step index from the end, default 2]"""
import sqlite3, sys, re

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
phys = [i for i, r in enumerate(rows) if "k_substep" in r[0] or "k_steps_resident" in r[0]]
# env steps = runs of physics kernels separated by > 300 us of other work
runs, cur_run = [], [phys[0]]
for a, b in zip(phys, phys[1:]):
    if rows[b][1] - rows[a][2] > 300e3:
        runs.append(cur_run); cur_run = []
    cur_run.append(b)
runs.append(cur_run)
if len(sys.argv) > 2 and sys.argv[2] == "all":     # one line per env step: physics, time until the next step, GPU idle in between, kernels in between
    for j in range(len(runs) - 1):
        a, b = runs[j], runs[j + 1]
        t1, tn = rows[a[-1]][2], rows[b[0]][1]
        bu, idle = t1, 0.0
        for n, s_, e_ in rows[a[-1] + 1:b[0]]:
            if s_ > bu: idle += s_ - bu
            bu = max(bu, e_)
        idle += max(0.0, tn - bu)
        comp = sum(1 for n, _, _ in rows[a[-1] + 1:b[0]] if "k_composite" in n)
        print(f"step {j:3d} (from end {len(runs) - 1 - j:3d}): physics {1e-6 * (t1 - rows[a[0]][1]):7.3f} ms, to next {1e-6 * (tn - t1):7.3f} ms, idle {1e-6 * idle:6.3f} ms, kernels between {b[0] - a[-1] - 1:4d}, composites {comp}")
    sys.exit(0)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
r0, r1 = runs[-k - 1], runs[-k]
t_phys0, t_phys1 = rows[r0[0]][1], rows[r0[-1]][2]
t_next = rows[r1[0]][1]
print(f"env step: physics {1e-6 * (t_phys1 - t_phys0):.3f} ms, then {1e-6 * (t_next - t_phys1):.3f} ms until the next step's first substep kernel")
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |rocprim::ROCPRIM_\d+_NS::detail::", "", n).split("(")[0][:60]
busy_until, idle, merged = t_phys1, 0.0, []
for n, s, e in rows[r0[-1] + 1:r1[0]]:
    if s > busy_until:
        gap = s - busy_until
        idle += gap
        if gap > 15e3:
            merged.append(("-- idle --", gap))
    merged.append((short(n), e - s))
    busy_until = max(busy_until, e)
if t_next > busy_until:
    idle += t_next - busy_until
    merged.append(("-- idle (before the next step's first kernel) --", t_next - busy_until))
print(f"GPU idle in between: {1e-6 * idle:.3f} ms")
for n, d in merged:
    if d > 8e3:
        print(f"  {d * 1e-3:8.1f} us  {n}")
