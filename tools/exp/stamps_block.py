#!/usr/bin/env python3
"""Block-level timeline of the LDS-DMA halo conv (C3_STAMPS build): kernel entry, setup done, loop start, loop end,
epilogue done -- for block 0 (first on its CU) and block 2048 (a later round)."""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madm_amd import ops
from madm_amd._lib import lib
cin, cout, hw, tile, gn = (int(x) for x in (sys.argv[1:6] + ["128", "128", "512", "9", "1"][len(sys.argv) - 1:]))
B = 2
x = torch.randn((B * hw * hw, cin), device="cuda").to(torch.bfloat16)
w = (torch.randn((cout, 9 * cin), device="cuda") / math.sqrt(9 * cin)).to(torch.bfloat16)
st = torch.zeros((B, cout, 2), dtype=torch.float64, device="cuda")
g = None
if gn:
    sums = torch.zeros((B, cin, 2), dtype=torch.float64, device="cuda")
    ops.groupnorm_stats(x, B, hw * hw, sums)
    g = ([sums], torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1, 32, 1e-5, True)
lib.madm_debug_set_conv_tile(tile)
for _ in range(3):
    ops.conv2d(x, w, B, hw, hw, N=cout, KH=3, KW=3, pad_t=1, pad_l=1, gn=g, splitk=1, stats=st)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); ops.conv2d(x, w, B, hw, hw, N=cout, KH=3, KW=3, pad_t=1, pad_l=1, gn=g, splitk=1, stats=st); e1.record(); torch.cuda.synchronize()
fn = ctypes.CDLL(os.environ["MADM_HIP_LIB"]).madm_debug_read_c3_stamps
buf = (ctypes.c_ulonglong * 24)()
assert fn(buf, 24) == 0
print(f"kernel {e0.elapsed_time(e1) * 1e3:.1f} us incl. event overhead")
for name, o in (("block 0", 0), ("block 2048", 8)):
    t = [buf[o + i] for i in range(5)]
    print(f"{name}: setup {t[1] - t[0]}  prologue(halo+fold+DMA) {t[2] - t[1]}  loop {t[3] - t[2]}  epilogue {t[4] - t[3]}  total {t[4] - t[0]} clocks")
print("block 2048 entry - block 0 entry:", buf[8] - buf[0], "clocks")
e = [buf[16 + i] for i in range(5)]
print(f"epilogue of block 0: stores+row epilogue {e[1] - e[0]}  shuffles {e[2] - e[1]}  barrier {e[3] - e[2]}  atomics issue {e[4] - e[3]}  drain (vmcnt 0) {buf[4] - e[4]}")
