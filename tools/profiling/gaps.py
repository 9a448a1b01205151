"""GPU idle gaps inside an env step from a rocprofv3 rocpd database: union of kernel intervals vs wall clock per step."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select start, end, {name} from kernels order by start").fetchall()
# find env-step boundaries: k_preprocess marks the start of each render
marks = [i for i, r in enumerate(rows) if "k_preprocess" in r[2]]
print("kernels", len(rows), "renders", len(marks))
for a, b in zip(marks[1:-1], marks[2:]):
    seg = rows[a:b]
    t0, t1 = seg[0][0], rows[b][0]
    busy, cur_end, gaps = 0, t0, []
    for s, e, n in seg:
        if s > cur_end:
            gaps.append((s - cur_end, n))
            cur_end = s
        if e > cur_end:
            busy += e - max(s, cur_end) if s < cur_end else e - s
            cur_end = e
    big = sorted(gaps, reverse=True)[:6]
    print(f"step wall {(t1-t0)/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(t1-t0-busy)/1e6:.3f} ms in {len(gaps)} gaps; largest (us, next kernel): " +
          ", ".join(f"{g/1e3:.0f}:{n.split('(')[0][-28:]}" for g, n in big))
