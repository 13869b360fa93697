#!/usr/bin/env python3
"""Block-level stamps of the register-staged igemm (GLDS_STAMPS build with the REG_BSTAMP patch, block 7):
entry, first loads, loop start, loop end, epilogue issued, stores drained.  Usage: hw cin cout k tile geglu(0/1)"""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madm_amd import ops
from madm_amd._lib import lib
hw, cin, cout, k, tile, geglu = (int(x) for x in sys.argv[1:7])
B = 2
x = torch.randn((B * hw * hw, cin), device="cuda").to(torch.bfloat16)
w = (torch.randn((cout, k * k * cin), device="cuda") / math.sqrt(k * k * cin)).to(torch.bfloat16)
bias = None if os.environ.get("NOBIAS") else torch.randn(cout, device="cuda")
lib.madm_debug_set_conv_tile(tile)
f = lambda: ops.conv2d(x, w, B, hw, hw, N=cout, KH=k, KW=k, pad_t=k // 2, pad_l=k // 2, bias=bias, splitk=1,
                       epilogue=ops.EPI_GEGLU if geglu else ops.EPI_NONE)
for _ in range(3):
    f()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
fn = ctypes.CDLL(os.environ["MADM_HIP_LIB"]).madm_debug_read_reg_stamps
buf = (ctypes.c_ulonglong * 16)()
assert fn(buf, 16) == 0
t = [buf[i] for i in range(6)]
print(f"M{B*hw*hw} N{cout} K{k*k*cin} tile {tile} geglu {geglu}: kernel {e0.elapsed_time(e1)*1e3:.1f} us (with event overhead)")
print(f"  setup {t[1]-t[0]}  prologue loads+store {t[2]-t[1]}  loop {t[3]-t[2]}  epilogue issue {t[4]-t[3]}  store drain {t[5]-t[4]}  total {t[5]-t[0]}")
e = [buf[i] for i in range(8, 14)]   # [8] = constants landed (+ LayerNorm fold), [13] = every store of the tile issued
print(f"  epilogue: entry -> constants {e[0]-t[3]}  all rows (adds, residual, stores issued) {e[5]-e[0]}  tail {t[4]-e[5]}")
