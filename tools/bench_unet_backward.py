#!/usr/bin/env python3
"""Times the UNet forward and backward.unet_backward (eager, HIP events around whole calls) at the bench workload's
size: bs = 2, 64x64 latent (512x512 images), taps (5, 8, 11), LoRA r = 8 on q / k / v / out.
python tools/bench_unet_backward.py [--reps 3] [--dtype bf16]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _LoraConfig:
    def __init__(self, r, lora_alpha):
        self.r, self.lora_alpha = r, lora_alpha
        self.init_lora_weights = "gaussian"
        self.target_modules = ["to_k", "to_q", "to_v", "to_out.0"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--mode", default="", help="fwd | lora | all: run only this leg (for rocprofv3 --kernel-trace)")
    args = ap.parse_args()
    from madm_amd import backward, ops, weights
    from madm_amd.nn import Tok
    from madm_amd.sd_unet import UNet2DConditionModel
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    B, hw, Lk, taps = 2, args.hw, 77, (5, 8, 11)
    unet = UNet2DConditionModel()
    weights.synth_init_(unet, 0, "unet.")
    unet.add_adapter(_LoraConfig(8, 8), "default")
    unet.set_adapter(["default"])
    with torch.no_grad():
        for n, p in unet.named_parameters():
            if ".lora_B." in n:
                p.normal_(0.0, 0.02)
    unet = unet.cuda()
    g = torch.Generator().manual_seed(1)
    kt = ops.k_tile(dtype)
    x = Tok(torch.randn((B * hw * hw, kt), generator=g).to(dtype).cuda(), B, hw, hw)
    ctx = (0.5 * torch.randn((B * Lk, 768), generator=g)).to(dtype).cuda()
    cond = (0.02 * torch.randn((B, 1280), generator=g)).cuda()
    ts = torch.full((B,), 60, dtype=torch.int64, device="cuda")
    _, feats = unet(x, ts, ctx, Lk, cond_emb=cond, unet_block_indices=taps)
    dtaps = [torch.randn(f.t.shape, generator=g).to(dtype).cuda() for f in feats]

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, max(e0.elapsed_time(e1), (time.perf_counter() - t0) * 1e3))
        return best

    if args.mode:
        fn = {"fwd": lambda: unet(x, ts, ctx, Lk, cond_emb=cond, unet_block_indices=taps),
              "lora": lambda: backward.unet_backward(unet, x, ts, ctx, Lk, dtaps, taps, cond_emb=cond, base_grads=False),
              "all": lambda: backward.unet_backward(unet, x, ts, ctx, Lk, dtaps, taps, cond_emb=cond, base_grads=True)}[args.mode]
        print(f"{args.mode}: {timed(fn):.1f} ms (eager, best of {args.reps})")
        return
    t_f = timed(lambda: unet(x, ts, ctx, Lk, cond_emb=cond, unet_block_indices=taps))
    t_l = timed(lambda: backward.unet_backward(unet, x, ts, ctx, Lk, dtaps, taps, cond_emb=cond, base_grads=False))
    t_a = timed(lambda: backward.unet_backward(unet, x, ts, ctx, Lk, dtaps, taps, cond_emb=cond, base_grads=True))
    print(f"UNet bs={B} latent {hw}x{hw} {args.dtype} (eager, host launch overhead included):")
    print(f"  forward                                   {t_f:8.1f} ms")
    print(f"  backward, LoRA mode (data grads + LoRA)   {t_l:8.1f} ms   (includes the recomputing forward)")
    print(f"  backward, all parameters                  {t_a:8.1f} ms   (includes the recomputing forward)")


if __name__ == "__main__":
    main()
