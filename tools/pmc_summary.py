#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per dispatch."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if want and want not in name:
            continue
        key = (name[:60], r.get("Grid_Size", r.get("Grid_Size_X", "")))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in sorted(acc.items()):
    print(key)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
