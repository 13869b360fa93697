#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs (tools/pmc.sh) per kernel name."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "."
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:60]
        if not re.search(pat, k):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k, v in agg.items():
    print(k)
    for c, val in sorted(v.items()):
        n = cnt[k][c]
        print(f"   {c:28s} {val / n:16.1f} per dispatch ({n} dispatches)")
