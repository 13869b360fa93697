"""Deterministic synthetic parameters and checkpoint loading.

SD-v1-4 weights cannot be downloaded in the build environment, so parity tests and benchmarks use
seeded synthetic parameters: every tensor is drawn from its own generator keyed by (seed, parameter
name), so any module tree with the same (diffusers) parameter names -- the HIP modules of this
package or the CPU oracle -- receives bit-identical values regardless of registration order.
  weights (dim >= 2): N(0, 1/fan_in)        norm scales: 1 + 0.1 N(0,1)
  biases / norm shifts: 0.1 N(0,1)          (non-trivial so that a dropped bias/scale is caught)
"""
import math
import os
import zlib

import torch


# MADM_SYNTH_CACHE=1 (tests/conftest.py): the drawn values are kept per (seed, name, shape), so the ~40 model constructions
# of a test session draw the 860 M synthetic parameters once (6.9 s per UNet on 8 cores otherwise); values are COPIED into
# the parameters, the cache is never aliased.  ~3.8 GB of host memory per seed.
_SYNTH_CACHE = {} if os.environ.get("MADM_SYNTH_CACHE") else None


def _gen(seed, name):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


@torch.no_grad()
def synth_init_(module, seed=0, prefix=""):
    """Fills every parameter of ``module`` in place (on its current device) and returns it."""
    for name, p in module.named_parameters():
        full = prefix + name
        key = (seed, full, tuple(p.shape))
        if _SYNTH_CACHE is not None and key in _SYNTH_CACHE:
            p.copy_(_SYNTH_CACHE[key].to(p.dtype))
            continue
        g = _gen(seed, full)
        leaf = name.rsplit(".", 1)[-1]
        if leaf in ("prompt_embed", "time_embed"):          # trunc_normal_(std=0.02) in the reference (ldm_base.py:653,671)
            v = 0.02 * torch.randn(p.shape, generator=g)
        elif leaf.startswith("alpha_"):                     # rand / zeros there; non-trivial gates here
            v = torch.rand(p.shape, generator=g)
        elif p.dim() >= 2:
            fan_in = p[0].numel()
            v = torch.randn(p.shape, generator=g) / math.sqrt(fan_in)
        elif name.endswith("weight"):   # GroupNorm / LayerNorm scale
            v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        else:
            v = 0.1 * torch.randn(p.shape, generator=g)
        if _SYNTH_CACHE is not None:
            _SYNTH_CACHE[key] = v
        p.copy_(v.to(p.dtype))
    return module


@torch.no_grad()
def synth_buffers_(module, seed=0, prefix=""):
    """Non-trivial BatchNorm running statistics (eval-mode head): mean 0.1 N(0,1), var 0.5 + U(0,1)."""
    for name, b in module.named_buffers():
        g = _gen(seed, prefix + name)
        if name.endswith("running_mean"):
            b.copy_(0.1 * torch.randn(b.shape, generator=g))
        elif name.endswith("running_var"):
            b.copy_(0.5 + torch.rand(b.shape, generator=g))
    return module


@torch.no_grad()
def randomize_lora_B_(module, seed=0):
    """LoRA B is zero at init (a no-op adapter); parity/bench runs give it N(0, 1/r) values."""
    for name, p in module.named_parameters():
        if ".lora_B." in name:
            r = p.shape[1]
            p.copy_(torch.randn(p.shape, generator=_gen(seed, name)) / r)
    return module


def load_diffusers_dir(module, path, subfolder):
    """Loads ``<path>/<subfolder>/diffusion_pytorch_model.{safetensors,bin}`` (the layout of the
    CompVis/stable-diffusion-v1-4 snapshot the reference reads, mtmadise_multi_lora.py:27-28)."""
    d = os.path.join(path, subfolder)
    st = os.path.join(d, "diffusion_pytorch_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(d, "diffusion_pytorch_model.bin"), map_location="cpu")
    # pre-0.25 VAE attention names -> current ones
    ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if "attentions" in parts and parts[-2] in ren and subfolder == "vae":
            parts[-2:-1] = ren[parts[-2]].split(".")
            k = ".".join(parts)
            if v.dim() == 4:
                v = v[:, :, 0, 0]
        out[k] = v.float()
    missing, unexpected = module.load_state_dict(out, strict=False)
    if missing or unexpected:
        raise RuntimeError(f"{subfolder}: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
    return module
