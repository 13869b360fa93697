#!/usr/bin/env python3
"""Round 6: the 16 x 16 halo conv's block timeline in REAL time (s_memrealtime, 100 MHz) next to the shader-clock counter (which
under-counts while the MFMA pipes are loaded), plus per-block (entry, end, CU, slot) records of ALL blocks: how long does a block
live, how long is a CU slot empty between two blocks, how many blocks does a slot run.
MADM_HIP_LIB=build/libmadm_hip_h16stamps.so DT=f16 python tools/exp/stamps_h16_rt.py [cin cout hw gn res]"""
import collections
import ctypes
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madm_amd import ops
from madm_amd._lib import lib

cin, cout, hw, gn, res = (int(x) for x in (sys.argv[1:6] + ["128", "128", "512", "0", "0"][len(sys.argv) - 1:]))
B = int(os.environ.get("B", "2"))
DT = torch.float16 if os.environ.get("DT", "f16") == "f16" else torch.bfloat16
x = torch.randn((B * hw * hw, cin), device="cuda").to(DT)
w = (torch.randn((cout, 9 * cin), device="cuda") / math.sqrt(9 * cin)).to(DT)
bias = torch.randn(cout, device="cuda")
resid = torch.randn((B * hw * hw, cout), device="cuda").to(DT) if res else None
g = None
st = torch.zeros((B, cout, 2), dtype=torch.float64, device="cuda")
if gn:
    sums = torch.zeros((B, cin, 2), dtype=torch.float64, device="cuda")
    ops.groupnorm_stats(x, B, hw * hw, sums)
    g = ([sums], torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1, 32, 1e-5, True)
lib.madm_debug_set_conv_tile(12)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.conv2d(x, w, B, hw, hw, N=cout, KH=3, KW=3, pad_t=1, pad_l=1, bias=bias, gn=g, splitk=1, stats=st, residual=resid)
torch.cuda.synchronize()
e0.record()
ops.conv2d(x, w, B, hw, hw, N=cout, KH=3, KW=3, pad_t=1, pad_l=1, bias=bias, gn=g, splitk=1, stats=st, residual=resid)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
NB = B * ((hw + 15) // 16) ** 2 * ((cout + 127) // 128)
n = 8192 + 4 * 8192
buf = (ctypes.c_ulonglong * n)()
lib.madm_debug_read_h16_stamps.restype = ctypes.c_int
assert lib.madm_debug_read_h16_stamps(buf, n) == 0
sc, rt = buf[0:4096], buf[4096:8192]
RT = 10.0   # ns per s_memrealtime tick


def d(a, b):   # (shader-clock ticks, real ns) between stamps a and b
    return sc[b] - sc[a], (rt[b] - rt[a]) * RT


print(f"cin {cin} cout {cout} {hw}x{hw} B {B} gn {gn} residual {res}: kernel {us:.1f} us by HIP events, {NB} blocks")
nck = cin // 64
tot = d(0, 5)
print(f"stamped block (wave 0): entry -> end of epilogue {tot[0]} shader ticks = {tot[1] / 1e3:.2f} us real "
      f"({tot[0] / tot[1]:.3f} ticks per ns overall)")
rows = [("set-up", 0, 1), ("first DMAs issued", 1, 2), ("GroupNorm fold", 2, 3)]
i = 8
taps_sc = taps_ns = tr_sc = tr_ns = wait_sc = wait_ns = 0
for ck in range(nck):
    a, b, c = i, i + 1, i + 2
    w_ = d(a, b); t_ = d(b, c)
    wait_sc += w_[0]; wait_ns += w_[1]; tr_sc += t_[0]; tr_ns += t_[1]
    i += 3
    first = i
    for tap in range(9):
        i += 4
    x_ = (sc[i - 1] - sc[c], (rt[i - 1] - rt[c]) * RT)
    taps_sc += x_[0]; taps_ns += x_[1]
    print(f"chunk {ck}: halo wait+barrier {w_[0]} ticks / {w_[1]:.0f} ns, transform {t_[0]} / {t_[1]:.0f} ns, 9 taps {x_[0]} ticks / {x_[1]:.0f} ns "
          f"= {x_[1] / 9:.0f} ns per tap ({x_[0] / max(x_[1], 1):.2f} ticks per ns in the tap loop)")
ep = d(4, 5)
pro = d(0, 3)
print(f"prologue {pro[0]} ticks / {pro[1]:.0f} ns; transforms {tr_sc} / {tr_ns:.0f} ns; tap loops {taps_sc} / {taps_ns:.0f} ns; "
      f"halo waits {wait_sc} / {wait_ns:.0f} ns; epilogue {ep[0]} / {ep[1]:.0f} ns")
print(f"shares of the block's real time: prologue {100 * pro[1] / tot[1]:.0f} %, transforms {100 * tr_ns / tot[1]:.0f} %, "
      f"tap loops {100 * taps_ns / tot[1]:.0f} %, epilogue {100 * ep[1] / tot[1]:.0f} %")
mf = 18.0 * nck / 2 * 64 * 16 if False else nck * 9 * 64 * 16
print(f"MFMA issue clocks of the wave: {nck * 9 * 64 * 16} (= {nck * 9 * 64 * 16 / 2.1e3:.1f} us at 2.1 GHz; the tap loops took {taps_ns / 1e3:.1f} us real)")

# ---- per-block records
recs = []
for b in range(min(NB, 8192)):
    s0, s1, hwid, la = buf[8192 + 4 * b:8192 + 4 * b + 4]
    if s1 <= s0:
        continue
    hw_lo, xcc = hwid & 0xffffffff, (hwid >> 32) & 0xf
    cu = (hw_lo >> 8) & 0xf
    sh = (hw_lo >> 12) & 0x1
    se = (hw_lo >> 13) & 0x7
    recs.append((s0, s1, (xcc, se, sh, cu), la & 0xfff, b))
t0 = min(r[0] for r in recs)
t1 = max(r[1] for r in recs)
life = sorted((r[1] - r[0]) * RT / 1e3 for r in recs)
print(f"{len(recs)} block records: first entry -> last end {(t1 - t0) * RT / 1e3:.1f} us; block lifetime us: min {life[0]:.1f} "
      f"p10 {life[len(life) // 10]:.1f} median {life[len(life) // 2]:.1f} p90 {life[9 * len(life) // 10]:.1f} max {life[-1]:.1f}")
slots = collections.defaultdict(list)
for r in recs:
    slots[(r[2], r[3])].append(r)
cus = collections.Counter(k[0] for k in slots)
print(f"{len(cus)} distinct (xcc, se, sh, cu) ids, {len(slots)} (cu, LDS base) slots; blocks per slot: "
      f"{dict(collections.Counter(len(v) for v in slots.values()))}")
gaps, starts, ends = [], [], []
for k, v in slots.items():
    v.sort()
    starts.append((v[0][0] - t0) * RT / 1e3)
    ends.append((t1 - v[-1][1]) * RT / 1e3)
    for a_, b_ in zip(v, v[1:]):
        gaps.append((b_[0] - a_[1]) * RT / 1e3)
gaps.sort()
if gaps:
    print(f"gap between two blocks of one slot (end of epilogue of wave 0 -> entry of the next block), us: min {gaps[0]:.2f} "
          f"median {gaps[len(gaps) // 2]:.2f} p90 {gaps[9 * len(gaps) // 10]:.2f} max {gaps[-1]:.2f}; sum per slot "
          f"{sum(gaps) / len(slots):.1f} us")
starts.sort(); ends.sort()
print(f"slot's first entry after the kernel's first: median {starts[len(starts) // 2]:.2f} us, max {starts[-1]:.2f}; idle tail of a slot before "
      f"the kernel's last end: median {ends[len(ends) // 2]:.2f} us, max {ends[-1]:.2f}")
# rounds: entries sorted
ent = sorted((r[0] - t0) * RT / 1e3 for r in recs)
q = [ent[int(len(ent) * f)] for f in (0.0, 0.25, 0.5, 0.75, 0.999)]
print("entry times of the blocks (us after the first), quantiles 0 / 25 / 50 / 75 / 100 %:", " ".join(f"{v:.1f}" for v in q))
