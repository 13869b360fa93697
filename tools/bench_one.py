#!/usr/bin/env python3
"""Runs ONE conv shape repeatedly (for rocprofv3 --pmc passes and quick A/B timing).
python tools/bench_one.py --M-hw 512 512 --B 2 --cin 128 --cout 128 --tile 4 [--gn] [--reps 20]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, nargs=2, default=[512, 512])
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--cin", type=int, default=128)
    ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--splitk", type=int, default=1)
    ap.add_argument("--gn", action="store_true")
    ap.add_argument("--residual", action="store_true")
    ap.add_argument("--rotate", type=int, default=1, help="cycle through this many weight buffers (cold weights, as in the model)")
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--check", type=int, default=0, help="compare the output (and fused statistics) with this tile code")
    args = ap.parse_args()
    from madm_amd import ops
    from madm_amd._lib import lib
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    H, W = args.hw
    B, Cin, Cout, k = args.B, args.cin, args.cout, args.k
    x = torch.randn((B * H * W, Cin), device="cuda").to(dtype)
    ws = [(torch.randn((Cout, k * k * Cin), device="cuda") / math.sqrt(k * k * Cin)).to(dtype) for _ in range(args.rotate)]
    it = [0]
    bias = torch.randn(Cout, device="cuda")
    gn = None
    if args.gn:
        sums = torch.zeros((B, Cin, 2), dtype=torch.float64, device="cuda")
        ops.groupnorm_stats(x, B, H * W, sums)
        gn = ([sums], torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1, 32, 1e-5, True)
    st = None if args.no_stats else torch.zeros((B, Cout, 2), dtype=torch.float64, device="cuda")
    lib.madm_debug_set_conv_tile(args.tile)
    res = torch.randn((B * H * W, Cout), device="cuda").to(dtype) if args.residual else None

    def run():
        it[0] += 1
        w = ws[it[0] % len(ws)]
        return ops.conv2d(x, w, B, H, W, N=Cout, KH=k, KW=k, pad_t=k // 2, pad_l=k // 2, bias=bias, stats=st, gn=gn, residual=res,
                          splitk=args.splitk)

    if args.check:
        if st is not None:
            st.zero_()
        out = run().float().clone()
        st_a = None if st is None else st.clone()
        lib.madm_debug_set_conv_tile(args.check)
        it[0] -= 1
        if st is not None:
            st.zero_()
        ref = run().float().clone()
        lib.madm_debug_set_conv_tile(args.tile)
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        serr = 0.0 if st is None else ((st_a - st).abs().max() / st.abs().max()).item()
        print(f"check vs tile {args.check}: max rel err {err:.3e}, stats rel err {serr:.3e}", "OK" if err < 2e-2 and serr < 1e-3 else "MISMATCH")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    if args.graph:   # no host launch overhead between the launches (small kernels)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(args.reps):
                run()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
    else:
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.reps * 1e3
    fl = 2.0 * B * H * W * Cout * k * k * Cin
    print(f"tile {args.tile} sk{args.splitk} rot{args.rotate} M{B * H * W} N{Cout} K{k * k * Cin}: {us:.1f} us  {fl / us / 1e6:.1f} TF/s")


if __name__ == "__main__":
    main()
