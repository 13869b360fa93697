#!/usr/bin/env python3
"""Per-launch timing of the MFMA kernels in one eager forward (HIP events on the launch stream):
which layer shapes the time goes to.  Usage: python tools/profile_layers.py [--dtype bf16] [--top 40]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--top", type=int, default=200)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=512)
    args = ap.parse_args()
    from madm_amd.ldm_rocm import LdmRocm
    from madm_amd import ops
    import bench
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    m = LdmRocm("", [], [5, 8, 11], [], input_range='-1+1', unet_block_indices_type='after', finetune_unet='no',
                compute_dtype=dtype, weights='synthetic', seed=0)
    inputs = bench.make_inputs(args.batch, args.size, torch.device("cuda"))
    for _ in range(2):
        m(inputs, "rgb")
    torch.cuda.synchronize()
    ops.PROFILE = []
    ops.PROFILE_DIFF = True          # [K] [K K] brackets: the difference is one launch without the event-pair cost
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    m(inputs, "rgb")
    e1.record()
    torch.cuda.synchronize()
    rec = ops.PROFILE
    ops.PROFILE = None
    ops.PROFILE_DIFF = False
    rows = {}
    for name, flops, a, b, desc, _, pr in rec:
        k = (name, desc)
        n, ms, fl = rows.get(k, (0, 0.0, 0.0))
        t = a.elapsed_time(b) if pr.e2 is None else max(b.elapsed_time(pr.e2) - a.elapsed_time(b), 1e-4)
        rows[k] = (n + 1, ms + t, fl + flops)
    tot = sum(v[1] for v in rows.values())
    print(f"eager forward {e0.elapsed_time(e1):.2f} ms; MFMA kernels {tot:.2f} ms in {len(rec)} launches")
    print(f"{'kernel':22s} {'shape':44s} {'n':>3s} {'ms':>8s} {'TF/s':>7s}")
    for (name, desc), (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f"{name:22s} {desc:44s} {n:3d} {ms:8.3f} {fl / ms / 1e9:7.1f}")


if __name__ == "__main__":
    main()
