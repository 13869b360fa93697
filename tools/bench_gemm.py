#!/usr/bin/env python3
"""The UNet's linear-layer GEMM shapes (bs = 2, 512 x 512) one by one: plain vs LayerNorm-folded, in a hipGraph with rotating
(cold) weights and inputs.  Same-box A/B of kernel builds: MADM_HIP_LIB=<other .so> python tools/bench_gemm.py ...
    python tools/bench_gemm.py [--dtype f16] [--reps 40] [--only geglu]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # name, M, K, N, geglu, residual, ln-foldable
    ("qkv_64", 8192, 320, 960, False, False, True),
    ("qkv_32", 2048, 640, 1920, False, False, True),
    ("qkv_16", 512, 1280, 3840, False, False, True),
    ("q_64", 8192, 320, 320, False, False, True),
    ("q_32", 2048, 640, 640, False, False, True),
    ("q_16", 512, 1280, 1280, False, False, True),
    ("out_64", 8192, 320, 320, False, True, False),
    ("out_32", 2048, 640, 640, False, True, False),
    ("out_16", 512, 1280, 1280, False, True, False),
    ("geglu_64", 8192, 320, 2560, True, False, True),
    ("geglu_32", 2048, 640, 5120, True, False, True),
    ("geglu_16", 512, 1280, 10240, True, False, True),
    ("ffout_64", 8192, 1280, 320, False, True, False),
    ("ffout_32", 2048, 2560, 640, False, True, False),
    ("ffout_16", 512, 5120, 1280, False, True, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--rotate", type=int, default=8)
    ap.add_argument("--only", default="")
    ap.add_argument("--tile", type=int, default=0)
    args = ap.parse_args()
    from madm_amd import ops, packing
    from madm_amd._lib import lib, LIB_PATH
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    kt = ops.k_tile(dtype)
    print("lib:", os.path.basename(LIB_PATH), "dtype", args.dtype)
    lib.madm_debug_set_conv_tile(args.tile)
    tot = {"plain": 0.0, "ln": 0.0}
    for name, M, K, N, geglu, resid, foldable in SHAPES:
        if args.only and args.only not in name:
            continue
        R = args.rotate
        xs = [torch.randn((M, K), device="cuda").to(dtype) for _ in range(R)]
        ocols = N // 2 if geglu else N
        res = torch.randn((M, ocols), device="cuda").to(dtype) if resid else None
        gamma, beta = torch.rand(K) + 0.5, 0.1 * torch.randn(K)
        ws, wl = [], []
        for _ in range(R):
            w = torch.randn((N, K)) / math.sqrt(K)
            b = torch.randn(N)
            ws.append((packing.pack_linear_weight(w, dtype, kt).cuda(), b.cuda()))
            wl.append(tuple(t.cuda() for t in packing.fold_layernorm(w, b, gamma, beta, dtype, kt)))
        out = torch.empty((M, ocols), device="cuda", dtype=dtype)
        epi = ops.EPI_GEGLU if geglu else ops.EPI_NONE
        row = f"{name:10s} M{M:5d} K{K:5d} N{N:6d}"
        for variant in ("plain", "ln"):
            if variant == "ln" and not foldable:
                continue
            it = [0]

            def run():
                it[0] += 1
                i = it[0] % R
                if variant == "plain":
                    ops.linear(xs[i], ws[i][0], bias=ws[i][1], residual=res, epilogue=epi, out=out)
                else:
                    ops.linear(xs[i], wl[i][0], bias=wl[i][1], residual=res, epilogue=epi, out=out, ln=(wl[i][2], 1e-5))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(args.reps):
                    run()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / args.reps * 1e3)
            tot[variant] += best
            row += f"   {variant}: {best:6.1f} us {2.0 * M * N * K / best / 1e6:6.0f} TF/s"
        print(row)
    print("sum of one launch each:", {k: round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
