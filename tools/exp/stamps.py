#!/usr/bin/env python3
"""Reads the in-kernel shader-clock stamps of the C3_STAMPS build of the halo conv (block 0 / wave 0):
per step: [0] step start, [1] MFMAs issued, [2] LDS stores issued, [3] after the barrier.
MADM_HIP_LIB=madm_amd/libmadm_hip_STAMPS.so python tools/exp/stamps.py [cin cout hw tile gn]"""
import ctypes
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madm_amd import ops
from madm_amd._lib import lib

cin, cout, hw, tile, gn = (int(x) for x in (sys.argv[1:6] + ["512", "512", "64", "5", "1"][len(sys.argv) - 1:]))
B = 2
x = torch.randn((B * hw * hw, cin), device="cuda").to(torch.bfloat16)
w = (torch.randn((cout, 9 * cin), device="cuda") / math.sqrt(9 * cin)).to(torch.bfloat16)
bias = torch.randn(cout, device="cuda")
g = None
if gn:
    sums = torch.zeros((B, cin, 2), dtype=torch.float64, device="cuda")
    ops.groupnorm_stats(x, B, hw * hw, sums)
    g = ([sums], torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1, 32, 1e-5, True)
lib.madm_debug_set_conv_tile(tile)
for _ in range(3):
    ops.conv2d(x, w, B, hw, hw, N=cout, KH=3, KW=3, pad_t=1, pad_l=1, bias=bias, gn=g, splitk=1)
torch.cuda.synchronize()
n = 9 * cin // 64
buf = (ctypes.c_ulonglong * (4 * n))()
lib.madm_debug_read_c3_stamps.restype = ctypes.c_int
rc = lib.madm_debug_read_c3_stamps(buf, 4 * n)
assert rc == 0, rc
st = [[buf[4 * s + k] for k in range(4)] for s in range(n)]
print(f"cin {cin} cout {cout} {hw}x{hw} tile {tile} gn {gn}: {n} steps, total {st[-1][3] - st[0][0]} clocks")
print("step:  load+mfma-issue   store-issue   barrier   | step total")
for s in range(min(n, 30)):
    a, b, c, d = st[s]
    print(f"{s:4d}: {b - a:8d} {c - b:8d} {d - c:8d}   | {d - a:8d}   gap to next {(st[s + 1][0] - d) if s + 1 < n else 0}")
tot = [sum(st[s][k + 1] - st[s][k] for s in range(n)) for k in range(3)]
print("sums:", tot, "per step avg", [t // n for t in tot])
