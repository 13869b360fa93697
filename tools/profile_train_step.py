#!/usr/bin/env python3
"""Per-launch timing of the profiled kernels (conv / linear forward + data gradient, weight gradient, attention) in one
eager training step of configs[3], HIP events around every launch (each pair costs a few microseconds: read the table for
the ORDER of the shapes and the rates of the long launches).  Usage: python tools/profile_train_step.py [--top 60]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=512)
    args = ap.parse_args()
    import bench
    from madm_amd import ops
    from madm_amd.train import MadmTrainer
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    dev = torch.device("cuda")
    model = bench.build_train_model(dtype, dev, False)
    trainer = MadmTrainer(model, lr=5e-6, weight_decay=0.05, grad_clip=0.01, dist=None, amp=True)
    data = bench.train_inputs(args.batch, args.size, dev)
    for _ in range(2):
        trainer.run_step(data)
    torch.cuda.synchronize()
    ops.PROFILE = []
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    trainer.run_step(data)
    e1.record()
    torch.cuda.synchronize()
    rec = ops.PROFILE
    ops.PROFILE = None
    rows, fam = {}, {}
    for name, flops, a, b, desc, nbytes, _ in rec:
        t = a.elapsed_time(b)
        n, ms, fl, by = rows.get((name, desc), (0, 0.0, 0.0, 0))
        rows[(name, desc)] = (n + 1, ms + t, fl + flops, by + nbytes)
        n, ms, fl = fam.get(name, (0, 0.0, 0.0))
        fam[name] = (n + 1, ms + t, fl + flops)
    tot = sum(v[1] for v in fam.values())
    print(f"eager step with events {e0.elapsed_time(e1):.1f} ms; profiled kernels {tot:.1f} ms in {len(rec)} launches")
    for name, (n, ms, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:28s} {n:5d} {ms:9.3f} ms {fl / ms / 1e9:8.1f} TF/s")
    print(f"{'kernel':26s} {'shape':46s} {'n':>3s} {'ms':>8s} {'TF/s':>7s} {'GB/s':>7s}")
    for (name, desc), (n, ms, fl, by) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f"{name:26s} {desc:46s} {n:3d} {ms:8.3f} {fl / ms / 1e9:7.1f} {by / ms / 1e6:7.0f}")


if __name__ == "__main__":
    main()
