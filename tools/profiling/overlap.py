"""Timeline of the physics kernels in a rocprofv3 rocpd database: how the two kernel chains of an env step overlap.
usage: overlap.py <results.db> [n_rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
extra = [c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols]
rows = cur.execute(f"select {name_col}, start, end {''.join(', ' + c for c in extra)} from kernels order by start").fetchall()
phys = [r for r in rows if "k_substep" in r[0] or "k_contact_finish" in r[0] or "k_self_finish" in r[0]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tail = phys[-(n + 700):-700] if len(phys) > n + 700 else phys[-n:]
t0 = tail[0][1]
for r in tail:
    short = "substep" if "k_substep" in r[0] else "finish "
    print(f"{short} {(r[1] - t0) / 1e3:9.2f} -> {(r[2] - t0) / 1e3:9.2f} us  ({(r[2] - r[1]) / 1e3:6.2f})  {r[3:]}")
# aggregate: busy time of the union vs the sum
iv = sorted((r[1], r[2]) for r in phys[-4000:])
union = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: union += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
union += ce - cs
print("last 4000 physics kernels: sum of durations", round(sum(e - s for s, e in iv) / 1e3, 1), "us, union", round(union / 1e3, 1), "us, span", round((iv[-1][1] - iv[0][0]) / 1e3, 1), "us")
