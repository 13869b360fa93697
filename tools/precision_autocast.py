#!/usr/bin/env python3
"""How far is the REFERENCE'S OWN arithmetic (fp16 CUDA autocast, fp16 VAE / frozen-UNet weights: engine/train_loop.py:277,
evaluation/evaluator.py:62-66, ldm_diffusers.py:248,253-255) from the fp32 oracle on the metric's path?  CPU only: the fp32
oracle and its autocast emulation (oracle/autocast_emul.py) on the same seeded inputs and weights; per-tensor relative L2
and max-relative error.  The HIP f16 mode's figures against the same fp32 oracle are in profiles/round*_precision_f16_bf16.txt.
    python tools/precision_autocast.py [--case full_t0|small_t0|small_t60|rect_t0] [--unet-weights f32|f16]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_all(case_name, unet_weights="f16", threads=None, variants=(False, True)):
    """-> {variant: (rows, seconds)} and the fp32 oracle's seconds; ONE model build and ONE fp32 reference run.
    variant: False / True = add_noise coefficients in fp16 (the reference) / fp32; ("wino", cins, min_hw[, operand dtype]) =
    the fp32-schedule variant with the VAE's eligible 3 x 3 convs as 16-bit Winograd F(2 x 2, 3 x 3) (oracle/autocast_emul.py)."""
    from golden_util import CASES, make_inputs
    from madm_amd import weights
    from oracle import sd_modules, ldm_path
    from oracle.autocast_emul import CudaAutocastF16, CudaAutocastF16Winograd, half_parameters_
    if threads:
        torch.set_num_threads(threads)
    case = CASES[case_name]
    images, cond, cond_emb, timesteps, noise = make_inputs(**case)
    vae = weights.synth_init_(sd_modules.AutoencoderKL().eval(), 0, "vae.")
    unet = weights.synth_init_(sd_modules.UNet2DConditionModel().eval(), 0, "unet.")
    sched = sd_modules.DDPMScheduler()
    t0 = time.time()
    with torch.no_grad():
        ref = ldm_path.ldm_forward(vae, unet, sched, noise, images, cond, cond_emb, timesteps=timesteps)
    t_ref = time.time() - t0
    half_parameters_(vae)                       # torch_dtype=torch.float16 (ldm_diffusers.py:248)
    if unet_weights == "f16":
        half_parameters_(unet)                  # frozen UNet: torch.float16 (:253); 'f32' = the fine-tuned UNet under autocast

    class _F32Schedule:
        # diffusers' add_noise casts alphas_cumprod to the latents' dtype (oracle/sd_modules.py DDPMScheduler.add_noise, the
        # restated diffusers 0.25 code): with fp16 latents sqrt(1 - fp16(0.99915)) = 0.03125 instead of 0.02915 at t = 0.
        # This variant keeps the latents fp32 through add_noise to show the rest of the arithmetic on its own.
        def add_noise(self, x, n, t):
            return sched.add_noise(x.float(), n, t)

    out = {}
    for f32_schedule in variants:
        t0 = time.time()
        mode = CudaAutocastF16()
        if isinstance(f32_schedule, tuple) and f32_schedule[0] == "fp8pv":
            mode.fp8_pv_min_keys = f32_schedule[1]
        elif isinstance(f32_schedule, tuple):
            mode = CudaAutocastF16Winograd(cins=f32_schedule[1], min_hw=f32_schedule[2],
                                           operand_dtype=f32_schedule[3] if len(f32_schedule) > 3 else torch.float16)
        with torch.no_grad(), mode:
            emu = ldm_path.ldm_forward(vae, unet, _F32Schedule() if f32_schedule else sched, noise, images, cond, cond_emb,
                                       timesteps=timesteps)
        if isinstance(f32_schedule, tuple) and f32_schedule[0] == "fp8pv":
            print(f"# {f32_schedule}: {mode.fp8_hits} self-attention calls ran their PV product in block-scaled e4m3", flush=True)
        elif isinstance(f32_schedule, tuple):
            print(f"# {f32_schedule}: {len(mode.hits)} convs ran as Winograd: {sorted(set(mode.hits))}", flush=True)
        rows = []
        pairs = [("latents", emu["latents"], ref["latents"]), ("sample", emu["sample"], ref["sample"])]
        pairs += [(f"tap{i}", a, b) for i, (a, b) in enumerate(zip(emu["unet_features"], ref["unet_features"]))]
        for name, a, b in pairs:
            a, b = a.float(), b.float()
            rows.append((name, tuple(b.shape), float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max())))
        out[f32_schedule] = (rows, time.time() - t0)
    return out, t_ref


def run(case_name, unet_weights="f16", threads=None, f32_schedule=False):
    out, t_ref = run_all(case_name, unet_weights, threads, variants=(f32_schedule,))
    rows, t_emu = out[f32_schedule]
    return rows, t_ref, t_emu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="full_t0")
    ap.add_argument("--unet-weights", default="f16", choices=["f16", "f32"])
    ap.add_argument("--winograd", action="store_true",
                    help="columns: fp32-schedule baseline; + Winograd F(2x2,3x3) on the VAE convs with 128 / 256 input channels at "
                         ">= 256 x 256 (the 16 x 16 halo kernel's layers); + on every 3 x 3 VAE conv; + the same with bf16 operands")
    ap.add_argument("--fp8-pv", action="store_true",
                    help="columns: fp32-schedule baseline; + the PV product of the 64 x 64 level's self-attention (4096 keys, the "
                         "five launches that are 0.45 ms of the step) in block-scaled e4m3; + of every self-attention (>= 64 keys)")
    args = ap.parse_args()
    if args.fp8_pv:
        variants = (True, ("fp8pv", 4096), ("fp8pv", 64))
        out, t_ref = run_all(args.case, args.unet_weights, variants=variants)
        print(f"case {args.case}: relative L2 (max rel) against the fp32 oracle; columns: {variants}")
        for i in range(len(out[True][0])):
            name = out[True][0][i][0]
            print(f"{name:10s} " + "   ".join(f"{out[v][0][i][2]:.3e} ({out[v][0][i][3]:.3e})" for v in variants))
        return
    if args.winograd:
        variants = (True, ("wino", (128, 256), 256), ("wino", (128, 256, 512), 64), ("wino", (128, 256), 256, torch.bfloat16))
        out, t_ref = run_all(args.case, args.unet_weights, variants=variants)
        print(f"case {args.case}: relative L2 (max rel) against the fp32 oracle; columns: {variants}")
        for i in range(len(out[True][0])):
            name = out[True][0][i][0]
            print(f"{name:10s} " + "   ".join(f"{out[v][0][i][2]:.3e} ({out[v][0][i][3]:.3e})" for v in variants))
        return
    out, t_ref = run_all(args.case, args.unet_weights)
    for f32_schedule in (False, True):
        rows, t_emu = out[f32_schedule]
        print(f"case {args.case}: fp32 oracle {t_ref:.1f} s, fp16-autocast emulation {t_emu:.1f} s on {torch.get_num_threads()} threads; "
              f"VAE weights f16, UNet weights {args.unet_weights}; add_noise coefficients "
              + ("in fp32 (NOT what the reference does: isolates the rest)" if f32_schedule else
                 "in the latents' fp16 as diffusers computes them (the reference's behaviour)"))
        print(f"{'tensor':10s} {'shape':22s} {'rel L2':>10s} {'max rel':>10s}   (emulated reference arithmetic vs fp32 oracle)")
        for name, shape, l2, mx in rows:
            print(f"{name:10s} {str(shape):22s} {l2:10.3e} {mx:10.3e}")


if __name__ == "__main__":
    main()
