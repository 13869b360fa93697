#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace --stats CSV directory: per-kernel calls / total / avg, per forward."""
import csv
import glob
import re
import sys

d = sys.argv[1]
nfwd = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{f}: total kernel time {tot / 1e6:.3f} ms over {nfwd:g} forwards = {tot / 1e6 / nfwd:.3f} ms/forward")
print(f"{'kernel':72s} {'calls/fwd':>9s} {'ms/fwd':>8s} {'avg_us':>9s} {'pct':>6s}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r["Name"])
    n = re.sub(r"void at::native::", "at::", n)
    print(f"{n[:72]:72s} {float(r['Calls']) / nfwd:9.1f} {float(r['TotalDurationNs']) / 1e6 / nfwd:8.3f} "
          f"{float(r['AverageNs']) / 1e3:9.2f} {float(r['Percentage']):6.2f}")
