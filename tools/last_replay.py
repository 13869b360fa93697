#!/usr/bin/env python3
"""Per-kernel totals of the LAST forward (graph replay) in a rocprofv3 --kernel-trace CSV.
usage: last_replay.py <kernel_trace.csv | dir> [anchor-substring=stem_conv3x3] [detail-regex]"""
import collections
import csv
import glob
import re
import sys

p = sys.argv[1]
if not p.endswith(".csv"):
    p = glob.glob(p + "/**/*kernel_trace.csv", recursive=True)[0]
anchor = sys.argv[2] if len(sys.argv) > 2 else "stem_conv3x3"
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
last = rows[idx[-1]:]
agg = collections.OrderedDict()
for r in last:
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:64]
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
print(f"{len(last)} kernels, sum {sum(v[1] for v in agg.values()):.1f} us, span {span:.1f} us")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:66s} {v[0]:4d} {v[1]:9.1f} us  avg {v[1] / v[0]:7.1f}")
if len(sys.argv) > 3:
    for r in last:
        if re.search(sys.argv[3], r["Kernel_Name"]):
            print(f"{re.sub(r'_ZN12_GLOBAL__N_1[0-9]+', '', r['Kernel_Name'])[:50]:52s} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d} x{r['Grid_Size_Y']:>3s} x{r['Grid_Size_Z']:>3s}"
                  f"  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us")
