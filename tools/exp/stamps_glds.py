#!/usr/bin/env python3
"""In-kernel stamps of the LDS-DMA igemm (GLDS_STAMPS build), block 0 / wave 0, per K-step:
[0] step start, [1] this tile's DMA landed (vmcnt), [2] barrier passed, [3] DMA issued + MFMAs issued.
MADM_HIP_LIB=madm_amd/libmadm_hip_STAMPS.so python tools/exp/stamps_glds.py M_hw cin cout k tile splitk"""
import ctypes
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madm_amd import ops
from madm_amd._lib import lib

hw, cin, cout, k, tile, sk = (int(x) for x in sys.argv[1:7])
B = 2
x = torch.randn((B * hw * hw, cin), device="cuda").to(torch.bfloat16)
ws = [(torch.randn((cout, k * k * cin), device="cuda") / math.sqrt(k * k * cin)).to(torch.bfloat16) for _ in range(8)]
lib.madm_debug_set_conv_tile(tile)
for i in range(9):
    ops.conv2d(x, ws[i % 8], B, hw, hw, N=cout, KH=k, KW=k, pad_t=k // 2, pad_l=k // 2, splitk=sk)
torch.cuda.synchronize()
n = (k * k * cin // 64) // sk
buf = (ctypes.c_ulonglong * 2048)()
fn = getattr(ctypes.CDLL(os.environ["MADM_HIP_LIB"]), "madm_debug_read_glds_stamps")
fn.restype = ctypes.c_int
assert fn(buf, 2048) == 0
st = [[buf[4 * s + j] for j in range(4)] for s in range(n)]
print(f"M{B * hw * hw} N{cout} K{k * k * cin} tile {tile} sk{sk}: {n} steps/block, total {st[-1][3] - st[0][0]} clocks")
print("step:  wait-DMA  barrier  issue+mfma | total   gap")
for s in range(n):
    a, b, c, d = st[s]
    print(f"{s:4d}: {b - a:8d} {c - b:8d} {d - c:8d} | {d - a:6d} {(st[s + 1][0] - d) if s + 1 < n else 0:6d}")
b = [buf[2000 + i] for i in range(5)]
print(f"block 0: setup {b[1] - b[0]}  prologue DMA issue {b[2] - b[1]}  loop {b[3] - b[2]}  epilogue (issue only) {b[4] - b[3]}  total {b[4] - b[0]}")
