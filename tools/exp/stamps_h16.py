#!/usr/bin/env python3
"""Reads the shader-clock stamps of the H16_STAMPS build of conv3x3_h16.hip (one block's wave 0).
MADM_HIP_LIB=madm_amd/libmadm_hip_H16STAMPS.so python tools/exp/stamps_h16.py [cin cout hw gn]"""
import ctypes
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madm_amd import ops
from madm_amd._lib import lib

LDS_ALLOC = "--lds-alloc" in sys.argv
if LDS_ALLOC:
    sys.argv.remove("--lds-alloc")
cin, cout, hw, gn, res = (int(x) for x in (sys.argv[1:6] + ["128", "128", "512", "0", "0"][len(sys.argv) - 1:]))
B = 2
DT = torch.float16 if os.environ.get("DT") == "f16" else torch.bfloat16
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
for _ in range(3):
    ops.conv2d(x, w, B, hw, hw, N=cout, KH=3, KW=3, pad_t=1, pad_l=1, bias=bias, gn=g, splitk=1, stats=st, residual=resid)
torch.cuda.synchronize()
nck = cin // 64
n = 8 + nck * (3 + 36)
buf = (ctypes.c_ulonglong * n)()
lib.madm_debug_read_h16_stamps.restype = ctypes.c_int
assert lib.madm_debug_read_h16_stamps(buf, n) == 0
t0 = buf[0]
if LDS_ALLOC:   # HW_REG_LDS_ALLOC as the first 1024 blocks read it: which values tell the CU's two resident workgroups apart?
    import collections
    al = (ctypes.c_ulonglong * 4024)()
    assert lib.madm_debug_read_h16_stamps(al, 4024) == 0
    hist = collections.Counter(int(v) & 0xffffffff for v in al[3000:4024])
    print("HW_REG_LDS_ALLOC values over blocks 0..1023:", ", ".join(f"{k:#010x} x {v}" for k, v in sorted(hist.items())))
    print("first 32 blocks:", " ".join(f"{int(v) & 0xffffffff:#x}" for v in al[3000:3032]))
print(f"cin {cin} cout {cout} {hw}x{hw} gn {gn} residual {res}: set-up {buf[1] - t0}, first DMAs issued +{buf[2] - buf[1]}, fold +{buf[3] - buf[2]}; "
      f"loop end at {buf[4] - t0}, epilogue {buf[5] - buf[4]}, total {buf[5] - t0}")
big = (ctypes.c_ulonglong * 2008)()
assert lib.madm_debug_read_h16_stamps(big, 2008) == 0
e = big[2000:2006]
print(f"epilogue (LDS-transposed): enter +{e[0] - buf[4]}, values -> LDS {e[1] - e[0]}, channel sums {e[2] - e[1]}, barrier {e[3] - e[2]}, "
      f"16-byte stores issued {e[4] - e[3]}, statistics atomics {e[5] - e[4]}")
i = 8
for ck in range(nck):
    a, b, c = buf[i], buf[i + 1], buf[i + 2]
    print(f"chunk {ck}: halo wait+barrier {b - a}, transform {c - b} (affine constants ready after {buf[6] - b} of the last chunk)")
    i += 3
    rows = []
    for tap in range(9):
        s0, s1, s2, s3 = buf[i:i + 4]
        prev = c if tap == 0 else buf[i - 1]
        rows.append((s0 - prev, s1 - s0, s2 - s1, s3 - s2))
        i += 4
    print("   tap: wait+barrier / DMA issue / first operands / 64 MFMAs issued")
    for tap, r in enumerate(rows):
        print(f"   {tap}: {r[0]:6d} {r[1]:6d} {r[2]:6d} {r[3]:6d}   = {sum(r)}")
