"""The DAFormer head's 1x1 convs at full resolution (M = B * 512 * 512 rows): which igemm tile per shape.
usage: python tools/exp/head_gemm.py [M]"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from madm_amd import ops, packing
from madm_amd._lib import lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
dt = torch.float16
g = torch.Generator(device="cuda").manual_seed(0)
for (K, N) in ((256, 1024), (1024, 256), (256, 128), (128, 128), (256, 256), (1024, 1024)):
    x = torch.randn((M, K), device="cuda", dtype=dt, generator=g)
    w = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5)
    wp = packing.pack_linear_weight(w.cpu(), dt, ops.k_tile(dt)).cuda()
    out = torch.empty((M, N), device="cuda", dtype=dt)
    gb = (x.numel() + out.numel()) * 2 / 1e9
    ref = None
    for tile in (0, 1, 2, 8, 14):
        lib.madm_debug_set_conv_tile(tile)
        try:
            ops.linear(x, wp, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.linear(x, wp, out=out)
            e1.record()
            torch.cuda.synchronize()
        finally:
            lib.madm_debug_set_conv_tile(0)
        ms = e0.elapsed_time(e1) / 5
        if ref is None:
            ref = out.clone()
        err = float((out.float() - ref.float()).abs().max())
        print(f"M{M} K{K} N{N} tile {tile:2d}: {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF/s {gb / ms:6.2f} TB/s  maxdiff {err:.1e}")
    del x, out
