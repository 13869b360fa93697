#!/usr/bin/env python3
"""Block timeline of the LDS-DMA 8 x 16 halo conv (-DC3D_STAMPS build of conv3x3_dma.hip), block 5 / thread 0.
usage: hw cin cout tile(9|10) [residual 0/1]"""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madm_amd import ops
from madm_amd._lib import lib
hw, cin, cout, tile = (int(x) for x in sys.argv[1:5])
res = int(sys.argv[5]) if len(sys.argv) > 5 else 0
B = 2
x = torch.randn((B * hw * hw, cin), device="cuda").to(torch.float16)
w = (torch.randn((cout, 9 * cin), device="cuda") / math.sqrt(9 * cin)).to(torch.float16)
bias = torch.randn(cout, device="cuda")
st = torch.zeros((B, cout, 2), dtype=torch.float64, device="cuda")
r = torch.randn((B * hw * hw, cout), device="cuda").to(torch.float16) if res else None
lib.madm_debug_set_conv_tile(tile)
f = lambda: ops.conv2d(x, w, B, hw, hw, N=cout, KH=3, KW=3, pad_t=1, pad_l=1, bias=bias, splitk=1, stats=st, residual=r)
for _ in range(3): f()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
fn = ctypes.CDLL(os.environ["MADM_HIP_LIB"]).madm_debug_read_c3d_stamps
buf = (ctypes.c_ulonglong * 256)()
assert fn(buf, 256) == 0
t = [buf[i] for i in range(7)]
print(f"M{B*hw*hw} N{cout} K{9*cin} tile {tile} residual {res}: kernel {e0.elapsed_time(e1)*1e3:.1f} us (with event overhead)")
print(f"  setup {t[1]-t[0]}  first halo + 2 weight tiles requested -> landed {t[2]-t[1]}  halo stored {t[3]-t[2]}  tap loop {t[4]-t[3]}  epilogue issue {t[5]-t[4]}  store drain {t[6]-t[5]}  total {t[6]-t[0]}")
nt = min(cin // 64, 3) * 9
print("  tap: wait-DMA barrier rest")
for k in range(nt):
    a, b_, c = (buf[16 + 3 * k + j] for j in range(3))
    nxt = buf[16 + 3 * (k + 1)] if k + 1 < nt else t[4]
    print(f"  {k:3d}: {b_-a:6d} {c-b_:6d} {nxt-c:6d}")
