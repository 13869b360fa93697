"""Colour augmentation of the mixed / target image inside the training step: ``strong_transform`` -> ``color_jitter`` +
``gaussian_blur`` (/root/reference/utils/dacs_transforms.py:11-78) on the HIP kernels of csrc/train.hip.

The reference builds ``kornia.augmentation.ColorJitter(brightness=s, contrast=s, saturation=s, hue=s)`` and
``kornia.filters.GaussianBlur2d((ky, kx), (sigma, sigma))`` per call; kornia is neither vendored nor pinned by the
reference (requirements.txt has no entry), so the arithmetic follows kornia's published 0.7 semantics as restated in
oracle/augment.py (parity of this piece is pinned to that restatement only):

* factors: brightness / contrast / saturation ~ U(1 - s, 1 + s) (lower bound clipped at 0), hue ~ U(-s, s) (|s| <= 0.5),
  the four transforms applied in a random order (``randperm(4)``); drawn from torch's RNG (``generator`` argument);
* brightness x * f, contrast x * f + mean(gray) * (1 - f), saturation (1 - f) * gray + f * x -- each clamped to [0, 1] --
  hue: HSV rotation by 2 pi f;
* blur: sigma ~ numpy U(0.15, 1.15) (dacs_transforms.py:62), kernel size floor(ceil(0.1 n) - 0.5 + ceil(0.1 n) % 2)
  (:63-70), separable normalised Gaussian, reflect border.

The branch conditions are the reference's: jitter when ``param['color_jitter'] > param['color_jitter_p']`` and the data has 3
channels (:43-45), blur when ``param['blur'] > 0.5`` (:60-61); ``mean`` / ``std`` are None in every shipped config
(pixel_mean = 0, cmdise.py:236-238)."""
import ctypes
import math

import numpy as np
import torch

from . import ops
from ._lib import lib


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def jitter_params(s, generator=None):
    """(brightness, contrast, hue, saturation factors, order) of one image, in kornia's sampling order."""
    def u(lo, hi):
        return float(lo + (hi - lo) * torch.rand((), generator=generator))
    if isinstance(s, dict):
        b, c, sat, h = (s.get(k, 0.0) for k in ("brightness", "contrast", "saturation", "hue"))
    else:
        b = c = sat = h = s
    fb = u(max(0.0, 1 - b), 1 + b)
    fc = u(max(0.0, 1 - c), 1 + c)
    fh = u(-min(h, 0.5), min(h, 0.5))
    fs = u(max(0.0, 1 - sat), 1 + sat)
    order = torch.randperm(4, generator=generator).tolist()
    return fb, fc, fh, fs, order


def color_jitter_image(img, fb, fc, fh, fs, order):
    """One RGB image [3, H, W] f32 in [0, 1] -> jittered copy."""
    ops._need_cuda(img)
    assert img.dtype == torch.float32 and img.is_contiguous() and img.shape[0] == 3
    HW = img.shape[1] * img.shape[2]
    out = img.clone()
    factors = (fb, fc, fs, 2 * math.pi * fh)       # transform index: 0 brightness, 1 contrast, 2 saturation, 3 hue
    for idx in order:
        gs = None
        if idx == 1:
            gs = torch.zeros(1, dtype=torch.float64, device=img.device)
            ops.check(lib.madm_gray_sum(out.data_ptr(), HW, gs.data_ptr(), _s()), "madm_gray_sum")
        ops.check(lib.madm_color_jitter_step(out.data_ptr(), out.data_ptr(), HW, int(idx), float(factors[idx]),
                                             None if gs is None else gs.data_ptr(), _s()), "madm_color_jitter_step")
    return out


def gaussian_kernel1d(ks, sigma, device):
    x = torch.arange(ks, dtype=torch.float32) - ks // 2
    if ks % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * sigma ** 2))
    return (g / g.sum()).to(device)


def blur_kernel_size(n):
    return int(np.floor(np.ceil(0.1 * n) - 0.5 + np.ceil(0.1 * n) % 2))


def gaussian_blur(data, sigma):
    """data [N, 3, H, W] f32 -> separable Gaussian blur, reflect border (kornia GaussianBlur2d)."""
    ops._need_cuda(data)
    N, C, H, W = data.shape
    data = data.float().contiguous()
    ky, kx = blur_kernel_size(H), blur_kernel_size(W)
    wy, wx = gaussian_kernel1d(ky, sigma, data.device), gaussian_kernel1d(kx, sigma, data.device)
    tmp, out = torch.empty_like(data), torch.empty_like(data)
    ops.check(lib.madm_blur_axis_f32(data.data_ptr(), tmp.data_ptr(), N * C, H, W, 1, kx, wx.data_ptr(), _s()), "madm_blur_axis_f32")
    ops.check(lib.madm_blur_axis_f32(tmp.data_ptr(), out.data_ptr(), N * C, H, W, 0, ky, wy.data_ptr(), _s()), "madm_blur_axis_f32")
    return out


def strong_color(param, data, generator=None, rng=np.random):
    """The colour part of ``strong_transform(param, data=...)`` (dacs_transforms.py:16-25): data [N, 3, H, W] in [0, 1]."""
    if data is None or data.shape[1] != 3:
        return data
    data = data.float().contiguous()
    if param['color_jitter'] > param['color_jitter_p']:
        assert param.get('mean') is None and param.get('std') is None, "denorm / renorm (pixel_mean != 0) is not built"
        outs = []
        for i in range(data.shape[0]):
            outs.append(color_jitter_image(data[i], *jitter_params(param['color_jitter_s'], generator)))
        data = torch.stack(outs)
    if param['blur'] > 0.5:
        sigma = rng.uniform(0.15, 1.15)
        data = gaussian_blur(data, sigma)
    return data
