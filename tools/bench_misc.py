#!/usr/bin/env python3
"""Times single non-MFMA kernels captured into a hipGraph (no host launch overhead between the launches).
python tools/bench_misc.py im2col|gn_apply|layernorm"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    from madm_amd import ops
    what = sys.argv[1] if len(sys.argv) > 1 else "im2col"
    if what == "im2col":
        img = torch.rand((2, 3, 512, 512), device="cuda")
        us = timed(lambda: ops.image_to_im2col3x3(img, torch.bfloat16, 64, 0.5, 0.5))
        print(f"im2col 2x3x512x512 -> [524288, 64] bf16: {us:.1f} us  ({67.1e6 / us / 1e6:.2f} TB/s written)")
        init = torch.tensor([float("inf"), float("-inf")], device="cuda")
        mm = torch.empty(2, device="cuda")

        def with_probe():
            mm.copy_(init)
            ops.image_to_im2col3x3(img, torch.bfloat16, 64, 0.5, 0.5, mm)
        us = timed(with_probe)
        print(f"  + range probe (reset + block min/max atomics): {us:.1f} us, minmax {mm.tolist()} "
              f"(expected {[(img.min().item() - 0.5) / 0.5, (img.max().item() - 0.5) / 0.5]})")
    elif what == "gn_apply":
        for (hw, c) in ((4096, 320), (1024, 640), (256, 1280), (65536, 128)):
            x = torch.randn((2 * hw, c), device="cuda").to(torch.bfloat16)
            st = torch.zeros((2, c, 2), dtype=torch.float64, device="cuda")
            ops.groupnorm_stats(x, 2, hw, st)
            g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
            us = timed(lambda: ops.groupnorm([x], 2, hw, 32, g, b, 1e-5, stats=[st], act=1))
            print(f"gn_apply B2 HW{hw} C{c}: {us:.1f} us")


if __name__ == "__main__":
    main()
