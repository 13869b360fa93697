#!/usr/bin/env python3
"""HBM-roofline measurement of the training-step tail kernels on the size of the fine-tuned parameter set
(UNet 859.5 M + projections/head/prompt 6.3 M fp32 parameters).  python tools/bench_optim.py [--n 865800000]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=865_800_000)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    from madm_amd._lib import lib
    n = args.n // 4 * 4
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 0.01
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda"); e = p.clone()
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    cases = [
        ("adamw_step", 28, lambda: lib.madm_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 5e-6, 0.9,
                                                       0.999, 1e-8, 0.05, 3, 1.0, s)),
        ("ema_update", 12, lambda: lib.madm_ema_update(e.data_ptr(), p.data_ptr(), n, 0.999, s)),
        ("grad_sumsq", 4, lambda: lib.madm_sumsq_f32(g.data_ptr(), n, out.data_ptr(), s)),
    ]
    for name, bpe, fn in cases:
        for _ in range(2):
            assert fn() == 0
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        gbs = n * bpe / ms / 1e6
        print(f"{name:12s} n={n}: {ms:8.3f} ms  {gbs:8.1f} GB/s algorithmic  ({gbs / 8000:.3f} of 8 TB/s spec, "
              f"{gbs / 6290:.3f} of 6.29 TB/s measured copy)")


if __name__ == "__main__":
    main()
