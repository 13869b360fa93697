#!/usr/bin/env python3
"""Determinism stress of the f32 eval forward (configs[2] graph): N forwards of the same image, every result compared bit for bit
with the first.  An intermittent kernel fault shows as a run that differs.  usage: stress_eval_f32.py [N] [dtype] [poison]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
POISON = len(sys.argv) > 3 and sys.argv[3] == "poison"     # fill every CU's LDS with NaNs in front of every forward
dtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
dev = torch.device("cuda")
model = bench.build_eval_model(dtype, dev)
ldm = model.backbone.feature_extractor.ldm_extractor
img = 255.0 * torch.rand((3, 512, 512), generator=torch.Generator().manual_seed(777))
call = [{"target_second_modality": img.to(dev)}]
with torch.no_grad():
    ref = model(call)[0]["sem_seg"].clone()
    lat0 = ldm.last_latents.clone()
    smp0 = ldm.last_sample.t.clone()
    bad = 0
    from madm_amd import ops
    for i in range(N):
        if POISON:
            ops.poison_lds(dev, 0x7fc00000 if i % 2 == 0 else 0x477fe000)     # NaN / 65504.0 (and the f16 pair 0x477f, 0xe000)
        out = model(call)[0]["sem_seg"]
        torch.cuda.synchronize()
        d_lat = int((ldm.last_latents != lat0).sum())
        d_smp = int((ldm.last_sample.t != smp0).sum())
        d_out = int((out != ref).sum())
        if d_lat or d_smp or d_out:
            bad += 1
            print(f"run {i}: latents differ in {d_lat} elements (max {float((ldm.last_latents - lat0).abs().max()):.3e}), UNet sample in "
                  f"{d_smp} (max {float((ldm.last_sample.t.float() - smp0.float()).abs().max()):.3e}), sem_seg in {d_out} "
                  f"(max {float((out - ref).abs().max()):.3e} of {float(ref.abs().max()):.3e})", flush=True)
print(f"{dtype}: {bad} of {N} runs differ from the first (library {os.environ.get('MADM_HIP_LIB', 'default')}"
      + (", LDS poisoned in front of every forward)" if POISON else ")"))
