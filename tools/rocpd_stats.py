#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (the default output of `rocprofv3 --kernel-trace --stats`)
into the per-kernel table the judge reads: calls, total / average / min / max duration, share of GPU time.
Usage: python tools/rocpd_stats.py gpurun_out/<dir>/<name>_results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, s, a, mn, mx in rows:
        short = n if len(n) < 110 else n[:107] + "..."
        lines.append(f"| `{short}` | {c} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/tot:.1f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
