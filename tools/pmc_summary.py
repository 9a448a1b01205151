#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc CSV passes (counter_collection.csv) per kernel: mean counter value per dispatch.
Usage: python tools/pmc_summary.py <dir with p1/ p2/ ...> [kernel substring filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if filt and filt not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
        n = len(next(iter(cs.values())))
        if n < 3 and not filt:
            continue
        print(f"== {k[:100]}  ({n} dispatches)")
        for c, v in sorted(cs.items()):
            print(f"   {c:28s} mean {sum(v)/len(v):16.1f}   min {min(v):14.1f}   max {max(v):14.1f}")


if __name__ == "__main__":
    main()
