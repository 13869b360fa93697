#!/usr/bin/env python3
"""Per-kernel totals of ONE forward (graph replay) in a rocprofv3 --kernel-trace CSV.

The window is the last forward that is delimited on BOTH sides: from the second-to-last launch of the anchor kernel (the VAE
stem, first kernel of a forward) up to, not including, the last one -- whatever the process runs after its last step
(calibration, probes, a second dtype) cannot leak into it.  The trailing window (last anchor to end of trace) is reported too
when it holds the same number of launches.  ``--expect N`` asserts the window's launch count (N +- 3).

usage: last_replay.py <kernel_trace.csv | dir> [anchor-substring=stem_conv3x3] [detail-regex] [--expect N]"""
import collections
import csv
import glob
import re
import sys

argv = list(sys.argv[1:])
expect = None
if "--expect" in argv:
    i = argv.index("--expect")
    expect = int(argv[i + 1])
    del argv[i:i + 2]
p = argv[0]
if not p.endswith(".csv"):
    p = glob.glob(p + "/**/*kernel_trace.csv", recursive=True)[0]
anchor = argv[1] if len(argv) > 1 else "stem_conv3x3"
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
assert len(idx) >= 2, f"need two launches of the anchor kernel '{anchor}', found {len(idx)}"
last = rows[idx[-2]:idx[-1]]
tail = rows[idx[-1]:]


def short(n):
    return re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)


agg = collections.OrderedDict()
for r in last:
    n = short(r["Kernel_Name"])[:64]
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
busy = sum(v[1] for v in agg.values())
print(f"{len(last)} kernels, sum {busy:.1f} us, span {span:.1f} us   (window: anchor launch {len(idx) - 1} of {len(idx)} up to the "
      f"next one; trailing window after the last anchor: {len(tail)} kernels"
      + (", same count)" if len(tail) == len(last) else " -- not a plain replay, ignored)"))
if expect is not None:
    assert abs(len(last) - expect) <= 3, f"expected {expect} +- 3 kernel launches per forward, found {len(last)}"
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:66s} {v[0]:4d} {v[1]:9.1f} us  avg {v[1] / v[0]:7.1f}")
if len(argv) > 2:
    for r in last:
        if re.search(argv[2], r["Kernel_Name"]):
            print(f"{short(r['Kernel_Name'])[:50]:52s} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d} x{r['Grid_Size_Y']:>3s} x{r['Grid_Size_Z']:>3s}"
                  f"  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us")
