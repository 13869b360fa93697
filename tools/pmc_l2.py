#!/usr/bin/env python3
"""L2 (TCC) hit rates per kernel AND grid from rocprofv3 --pmc passes (tools/exp/r4_evidence1.sh): one kernel template serves
many layer shapes, the grid tells them apart.  usage: pmc_l2.py <dir with pmc_*/ subdirectories>"""
import collections
import csv
import glob
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:56]
        wg = int(r.get("Workgroup_Size") or 256)
        grid = int(r.get("Grid_Size") or 0) // max(wg, 1)
        key = (k, grid)
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[key][r["Counter_Name"]].add(r["Dispatch_Id"])


def mean(key, c):
    n = len(disp[key][c])
    return agg[key][c] / n if n else 0.0


print(f"{'kernel':58s} {'WGs':>7s} {'n':>4s} {'TCC hit':>12s} {'TCC miss':>12s} {'hit rate':>8s} {'TCC req':>12s} {'TCP->TCC rd':>12s} {'TCP access':>12s}")
for key in sorted(agg, key=lambda k: -(mean(k, "TCC_HIT_sum") + mean(k, "TCC_MISS_sum")) * len(disp[k]["TCC_HIT_sum"])):
    h, m = mean(key, "TCC_HIT_sum"), mean(key, "TCC_MISS_sum")
    if h + m == 0:
        continue
    print(f"{key[0]:58s} {key[1]:7d} {len(disp[key]['TCC_HIT_sum']):4d} {h:12.0f} {m:12.0f} {h / (h + m):8.3f} {mean(key, 'TCC_REQ_sum'):12.0f} "
          f"{mean(key, 'TCP_TCC_READ_REQ_sum'):12.0f} {mean(key, 'TCP_TOTAL_CACHE_ACCESSES_sum'):12.0f}")
