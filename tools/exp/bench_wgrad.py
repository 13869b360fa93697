"""conv2d_wgrad on the training step's shapes (profiles/round3_train_step_layers.txt): us, TFLOP/s."""
import math, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madm_amd import ops
dt = torch.float16
def run(B, H, Cin, N, k, reps=3):
    M = B * H * H
    x = torch.randn((M, Cin), device="cuda").to(dt)
    dout = torch.randn((M, N), device="cuda").to(dt)
    f = lambda: ops.conv2d_wgrad(x, dout, B, H, H, KH=k, KW=k, pad_t=k // 2, pad_l=k // 2)
    for _ in range(2): dw = f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): dw = f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    K = k * k * Cin
    print(f"M{M} N{N} K{K} k{k}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s  (incl. the zero fill of dw)  checksum {float(dw.abs().sum()):.6e}")
run(2, 512, 1024, 256, 3)
run(2, 512, 1024, 256, 1)
run(2, 512, 128, 128, 3)
run(2, 64, 320, 320, 3)
run(2, 32, 640, 640, 1, 10)
run(2, 16, 1280, 1280, 1, 10)
run(2, 64, 320, 320, 1, 10)
run(2, 16, 1280, 1280, 3, 10)
run(2, 8, 1280, 1280, 3, 10)
