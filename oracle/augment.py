"""ORACLE (test infrastructure only): torch-CPU restatement of the colour augmentation inside ``strong_transform``
(/root/reference/utils/dacs_transforms.py:40-78), i.e. of ``kornia.augmentation.ColorJitter`` and
``kornia.filters.GaussianBlur2d``.

PARITY UNPINNED: kornia is a third-party dependency that the reference neither vendors nor pins (no requirements.txt entry;
it arrives transitively) and that is absent from this image, so there is no reference output to check against.  The
functions restate kornia 0.7's published algorithm (enhance/adjust.py: adjust_brightness_accumulative,
adjust_contrast_with_mean_subtraction, adjust_saturation_with_gray_subtraction, adjust_hue; color/hsv.py; filters/gaussian.py,
filters/kernels.py:get_gaussian_kernel1d; filter2d_separable with border_type='reflect') and anchor on the reference's call
sites: ColorJitter(brightness=s, contrast=s, saturation=s, hue=s) with s = 0.2 and GaussianBlur2d(kernel_size, (sigma,
sigma)) (dacs_transforms.py:47-52,62-75)."""
import math

import torch
import torch.nn.functional as F


def rgb_to_grayscale(img):
    r, g, b = img.unbind(-3)
    return (0.299 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)


def rgb_to_hsv(image, eps=1e-8):
    max_rgb, argmax_rgb = image.max(-3)
    min_rgb = image.min(-3)[0]
    deltac = max_rgb - min_rgb
    v = max_rgb
    s = deltac / (max_rgb + eps)
    deltac = torch.where(deltac == 0, torch.ones_like(deltac), deltac)
    rc, gc, bc = torch.unbind(max_rgb.unsqueeze(-3) - image, dim=-3)
    h = torch.stack([bc - gc, (rc - bc) + 2.0 * deltac, (gc - rc) + 4.0 * deltac], dim=-3) / deltac.unsqueeze(-3)
    h = torch.gather(h, dim=-3, index=argmax_rgb.unsqueeze(-3)).squeeze(-3)
    h = (h / 6.0) % 1.0
    return torch.stack([2 * math.pi * h, s, v], dim=-3)


def hsv_to_rgb(image):
    h = image[..., 0, :, :] / (2 * math.pi)
    s, v = image[..., 1, :, :], image[..., 2, :, :]
    hi = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - hi
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    hi = hi.long()
    idx = torch.stack([hi, hi + 6, hi + 12], dim=-3)
    out = torch.stack((v, q, p, p, t, v, t, v, v, q, p, p, p, p, t, v, v, q), dim=-3)
    return torch.gather(out, -3, idx)


def color_jitter_image(img, fb, fc, fh, fs, order):
    """img [3, H, W] in [0, 1]; transform index 0 brightness, 1 contrast, 2 saturation, 3 hue (kornia ColorJitter)."""
    x = img.clone()
    for idx in order:
        if idx == 0:
            x = torch.clamp(x * fb, 0, 1)
        elif idx == 1:
            m = rgb_to_grayscale(x).mean((-2, -1), True)
            x = torch.clamp(x * fc + m * (1 - fc), 0, 1)
        elif idx == 2:
            x = torch.clamp((1 - fs) * rgb_to_grayscale(x) + fs * x, 0, 1)
        else:
            hsv = rgb_to_hsv(x)
            h = torch.fmod(hsv[0] + 2 * math.pi * fh, 2 * math.pi)
            x = hsv_to_rgb(torch.stack([h, hsv[1], hsv[2]], dim=0))
    return x


def gaussian_kernel1d(ks, sigma):
    x = torch.arange(ks, dtype=torch.float32) - ks // 2
    if ks % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * sigma ** 2))
    return g / g.sum()


def gaussian_blur(data, ky, kx, sigma):
    """data [N, C, H, W]; separable, reflect border."""
    N, C, H, W = data.shape
    wy, wx = gaussian_kernel1d(ky, sigma), gaussian_kernel1d(kx, sigma)
    x = F.pad(data, (kx // 2, kx // 2, 0, 0), mode='reflect')
    x = F.conv2d(x.reshape(N * C, 1, H, -1), wx.view(1, 1, 1, kx))
    x = F.pad(x, (0, 0, ky // 2, ky // 2), mode='reflect')
    x = F.conv2d(x, wy.view(1, 1, ky, 1))
    return x.reshape(N, C, H, W)
