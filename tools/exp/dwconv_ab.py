"""A/B of the depthwise dilated 3x3 kernels on the head's tensor (DAFormer SepASPP: C channels at the feature size).
usage: python tools/exp/dwconv_ab.py [B] [H] [C]; MADM_DWCONV_KERNEL = 1 plain / 2 comb / 3 column walk is set per call."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from madm_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
W = H
dt = torch.float16
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((B * H * W, C), device="cuda", dtype=dt, generator=g)
w = torch.randn((9, C), device="cuda", generator=g) / 3
s = 1 + 0.1 * torch.randn((C,), device="cuda", generator=g)
t = 0.1 * torch.randn((C,), device="cuda", generator=g)
gb = 2 * x.numel() * x.element_size() / 1e9
for dil in (6, 12, 18):
    outs = {}
    for k in (2, 3):
        os.environ["MADM_DWCONV_KERNEL"] = str(k)
        y = ops.dwconv3x3(x, w, s, t, B, H, W, dil)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.dwconv3x3(x, w, s, t, B, H, W, dil, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        outs[k] = (y.clone(), ms)
        print(f"B{B} {H}x{W} C{C} dil {dil} kernel {k}: {ms * 1e3:8.1f} us  {gb / ms:6.2f} TB/s (read + write once)")
    print("   bit-identical:", torch.equal(outs[2][0], outs[3][0]))

