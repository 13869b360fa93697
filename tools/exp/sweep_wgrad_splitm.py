"""Slice count (grid.z) of conv2d_wgrad on the UNet's small layers: us per launch (incl. the zero fill of dw)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madm_amd import ops
dt = torch.float16
def run(B, H, Cin, N, k):
    M = B * H * H
    x = torch.randn((M, Cin), device="cuda").to(dt); dout = torch.randn((M, N), device="cuda").to(dt)
    res = []
    for sm in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
        f = lambda: ops.conv2d_wgrad(x, dout, B, H, H, KH=k, KW=k, pad_t=k // 2, pad_l=k // 2, splitm=sm)
        try:
            for _ in range(2): f()
        except Exception as e:
            continue
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        res.append(f"{sm}:{e0.elapsed_time(e1) / 10 * 1e3:.1f}")
    print(f"M{M} N{N} K{k*k*Cin} k{k}: " + "  ".join(res))
for cfg in ((2, 64, 320, 320, 1), (2, 32, 640, 640, 1), (2, 16, 1280, 1280, 1), (2, 8, 1280, 1280, 1), (2, 64, 320, 320, 3), (2, 32, 640, 640, 3),
            (2, 16, 1280, 1280, 3), (2, 8, 1280, 1280, 3), (2, 16, 1280, 10240, 1), (2, 64, 320, 2560, 1), (2, 16, 5120, 1280, 1), (2, 32, 2560, 640, 1)):
    run(*cfg)
