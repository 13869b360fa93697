#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 --kernel-trace CSV of the staged pipeline: per queue busy time, time with k kernels in
flight, and for the heavy kernel classes the share of their run time during which another queue's kernel also ran.
FINDING (round 3): under --kernel-trace the queues do not overlap at all -- 96.6 % of the pipelined window has exactly ONE
kernel in flight and a step takes 22 ms instead of 6.2: the profiler serialises dispatches, so the pipeline's overlap cannot be
read off a kernel trace (use tools/exp/skip_sensitivity.sh instead).
usage: pipeline_trace.py <dir | kernel_trace.csv>"""
import collections
import csv
import glob
import re
import sys

p = sys.argv[1]
if not p.endswith(".csv"):
    p = glob.glob(p + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(p))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region: the last 60 % of the trace (warm-up, captures and the serial reference come first / last)
t_lo = int(rows[0]["Start_Timestamp"])
t_hi = int(rows[-1]["End_Timestamp"])
qs = collections.Counter(r["Queue_Id"] for r in rows)
print("queues (kernels):", dict(qs))
# steady-state window: find the longest stretch in which >= 3 queues are active within every 2 ms
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1, r))
    ev.append((int(r["End_Timestamp"]), -1, r))
ev.sort(key=lambda e: (e[0], e[1]))
# windows of 1 ms: number of distinct queues seen
W = 1_000_000
buckets = collections.defaultdict(set)
for r in rows:
    for b in range(int(r["Start_Timestamp"]) // W, int(r["End_Timestamp"]) // W + 1):
        buckets[b].add(r["Queue_Id"])
good = sorted(b for b, s in buckets.items() if len(s) >= 3)
# longest run of consecutive good buckets
best, cur = (0, 0), None
for b in good:
    if cur is None or b != cur[1] + 1:
        cur = [b, b]
    else:
        cur[1] = b
    if cur[1] - cur[0] > best[1] - best[0]:
        best = (cur[0], cur[1])
lo, hi = best[0] * W, (best[1] + 1) * W
sel = [r for r in rows if int(r["Start_Timestamp"]) >= lo and int(r["End_Timestamp"]) <= hi]
span = (hi - lo) / 1e6
print(f"steady window: {span:.1f} ms, {len(sel)} kernels")
# time with k kernels in flight
active, last, hist = 0, lo, collections.Counter()
for t, d, r in ev:
    if t < lo or t > hi:
        continue
    hist[active] += t - last
    last = t
    active += d
tot = sum(hist.values())
print("kernels in flight -> share of the window:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
busy = collections.Counter()
for r in sel:
    busy[r["Queue_Id"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("busy share per queue:", {q: round(v / (hi - lo), 3) for q, v in busy.items()})


def cls(n):
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    for k in ("conv3x3_h16", "conv3x3_halo_dma", "igemm_glds_kernel", "igemm_kernel", "igemm_apanel", "attn_kernel", "gn_apply", "splitk_reduce"):
        if k in n:
            return k
    return "other"


# per class: own time, and time during which at least one kernel of ANOTHER queue was running
iv = collections.defaultdict(list)
for r in sel:
    iv[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
own, shared = collections.Counter(), collections.Counter()
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    c = cls(r["Kernel_Name"])
    own[c] += e - s
    covered = []
    for q, lst in iv.items():
        if q == r["Queue_Id"]:
            continue
        for a, b in lst:
            if b > s and a < e:
                covered.append((max(a, s), min(b, e)))
    covered.sort()
    u, cs_, ce = 0, None, None
    for a, b in covered:
        if cs_ is None:
            cs_, ce = a, b
        elif a <= ce:
            ce = max(ce, b)
        else:
            u += ce - cs_
            cs_, ce = a, b
    if cs_ is not None:
        u += ce - cs_
    shared[c] += u
print(f"{'class':22s} {'ms in window':>12s} {'per step':>9s} {'beside another queue':>22s}")
steps = span / 6.2
for c, v in sorted(own.items(), key=lambda kv: -kv[1]):
    print(f"{c:22s} {v / 1e6:12.2f} {v / 1e6 / steps:9.3f} {shared[c] / v:22.2f}")
print(f"sum of kernel time / window = {sum(own.values()) / (hi - lo):.2f} (kernels in flight on average)")
