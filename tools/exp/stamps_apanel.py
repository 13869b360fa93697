#!/usr/bin/env python3
"""Shader-clock timeline of wave 0 / workgroup 0 of the A-stationary GEMM (igemm_apanel.hip built with -DAP_STAMPS into
build/libmadm_APSTAMPS.so):  MADM_HIP_LIB=build/libmadm_APSTAMPS.so python tools/exp/stamps_apanel.py [M K N]"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madm_amd import ops, packing            # noqa: E402
from madm_amd._lib import lib                # noqa: E402

M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 320, 2560)
dtype = torch.float16
x = torch.randn((M, K), device="cuda").to(dtype)
w = packing.pack_linear_weight(torch.randn((N, K)) / math.sqrt(K), dtype, 64).cuda()
b = torch.randn(N).cuda()
lib.madm_debug_set_conv_tile(13)
for _ in range(3):
    ops.linear(x, w, bias=b, epilogue=ops.EPI_GEGLU)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 1024)()
lib.madm_debug_read_ap_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.madm_debug_read_ap_stamps(buf, 1024)
t = list(buf)
base = t[0]
print("prologue: issue %d  wait %d  barrier %d  (ln) %d" % (t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3]))
print("total %d clocks" % (t[5] - t[0]))
S = 0
while 8 + S * 4 + 3 < 1024 and t[8 + S * 4 + 3] > t[0]:
    S += 1
prev = t[4]
rows = []
for s in range(S):
    a, b_, c, d = t[8 + s * 4: 12 + s * 4]
    rows.append((a - prev, b_ - a, c - b_, d - c))
    prev = d
for s, r in enumerate(rows[:24]):
    print("step %2d: gap(epilogue) %5d  vmwait %5d  dma issue %4d  reads+mfma %5d" % ((s,) + r))
n = len(rows)
print("steps %d: mean gap %.0f vmwait %.0f dma %.0f compute %.0f" % (n, *(sum(r[i] for r in rows) / n for i in range(4))))
