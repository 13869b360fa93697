#!/usr/bin/env python3
"""Times the backward building blocks (weight gradient, data gradient through the forward kernel, weight repack) on
the conv / linear shapes of the bs = 2 / 512x512 forward, inside a hipGraph (no host launch overhead), next to the
forward conv of the same shape.  python tools/bench_backward.py [--reps 10] [--dtype bf16] [--splitm 0]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# name, B, H, W, Cin, Cout, k
SHAPES = [
    ("vae 128->128 @512", 2, 512, 512, 128, 128, 3),
    ("vae 256->256 @256", 2, 256, 256, 256, 256, 3),
    ("vae 512->512 @128", 2, 128, 128, 512, 512, 3),
    ("vae 512->512 @64", 2, 64, 64, 512, 512, 3),
    ("unet 320->320 @64", 2, 64, 64, 320, 320, 3),
    ("unet 640->640 @32", 2, 32, 32, 640, 640, 3),
    ("unet 1280->1280 @16", 2, 16, 16, 1280, 1280, 3),
    ("unet 1280->1280 @8", 2, 8, 8, 1280, 1280, 3),
    ("attn proj 320 (L 4096)", 2, 4096, 1, 320, 320, 1),
    ("ff geglu 320->2560", 2, 4096, 1, 320, 2560, 1),
    ("ff out 1280->320", 2, 4096, 1, 1280, 320, 1),
    ("attn proj 1280 (L 256)", 2, 256, 1, 1280, 1280, 1),
    ("ff geglu 1280->10240", 2, 256, 1, 1280, 10240, 1),
    ("head 1x1 1024->256 @512", 2, 512 * 512, 1, 1024, 256, 1),
    ("head 3x3 1280->256 @512", 2, 512, 512, 1280, 256, 3),
]


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s)
        g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--splitm", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--sweep", action="store_true", help="weight gradient only, over a list of pixel-slice counts")
    args = ap.parse_args()
    from madm_amd import ops
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    print(f"{'shape':28s} {'GFLOP':>8s} | {'fwd us':>8s} {'TF/s':>6s} | {'wgrad us':>8s} {'TF/s':>6s} | {'dgrad us':>8s} {'TF/s':>6s} | {'repack us':>9s}")
    for name, B, H, W, Cin, Cout, k in SHAPES:
        if args.only and args.only not in name:
            continue
        M = B * H * W
        x = torch.randn((M, Cin), device="cuda").to(dtype)
        dy = torch.randn((M, Cout), device="cuda").to(dtype)
        w = (torch.randn((Cout, k * k * Cin), device="cuda") / math.sqrt(k * k * Cin)).to(dtype)
        dw = torch.zeros((Cout, k * k * Cin), device="cuda")
        wt = ops.pack_dgrad_weights(w, k * k)
        fl = 2.0 * M * Cout * k * k * Cin
        if args.sweep:
            steps = (M + 31) // 32 if dtype == torch.bfloat16 else (M + 15) // 16
            tiles = ((Cout + 127) // 128) * ((k * k * Cin + 127) // 128)
            cands = sorted({c for c in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, max(1, 512 // tiles),
                                        max(1, 256 // tiles), max(1, 1024 // tiles)) if c <= steps})
            row = []
            for sm in cands:
                t = timed(lambda: ops.conv2d_wgrad(x, dy, B, H, W, KH=k, KW=k, pad_t=k // 2, pad_l=k // 2, dw=dw,
                                                   splitm=sm), args.reps)
                row.append((t, sm))
            t0 = timed(lambda: ops.conv2d_wgrad(x, dy, B, H, W, KH=k, KW=k, pad_t=k // 2, pad_l=k // 2, dw=dw), args.reps)
            best = min(row)
            print(f"{name:28s} tiles {tiles:4d} steps {steps:6d} auto {t0:7.1f} best {best[0]:7.1f} @ {best[1]:3d} | "
                  + " ".join(f"{sm}:{t:.0f}" for t, sm in row), flush=True)
            continue
        t_f = timed(lambda: ops.conv2d(x, w, B, H, W, N=Cout, KH=k, KW=k, pad_t=k // 2, pad_l=k // 2), args.reps)
        t_w = timed(lambda: ops.conv2d_wgrad(x, dy, B, H, W, KH=k, KW=k, pad_t=k // 2, pad_l=k // 2, dw=dw,
                                             splitm=args.splitm), args.reps)
        t_d = timed(lambda: ops.conv2d_dgrad(dy, wt, B, H, W, C=Cin, KH=k, KW=k, pad_t=k // 2, pad_l=k // 2), args.reps)
        t_p = timed(lambda: ops.pack_dgrad_weights(w, k * k), args.reps)
        print(f"{name:28s} {fl / 1e9:8.1f} | {t_f:8.1f} {fl / t_f / 1e6:6.0f} | {t_w:8.1f} {fl / t_w / 1e6:6.0f} | "
              f"{t_d:8.1f} {fl / t_d / 1e6:6.0f} | {t_p:9.1f}", flush=True)


if __name__ == "__main__":
    main()
