"""Label / pseudo-label pipeline of the self-training step on the device.

Host-side mirror of what ``MTMADISE.forward`` does around the hot path, with the reference's names and argument
meaning, running on the HIP kernels of ``csrc/labels.hip`` instead of PIL / numpy / host syncs:

* ``convert_label_to_rgb``  -- /root/reference/modeling/meta_arch/mtmadise.py:159-175
* ``pseudo_labels``         -- mtmadise.py:339-349 (bilinear up-sampling, softmax, max, threshold, pseudo weight)
* ``get_class_masks`` / ``one_mix`` / ``class_mix`` -- /root/reference/utils/dacs_transforms.py:81-111 (ClassMix)

No CPU fallback: every function raises on CPU tensors.
"""
import ctypes

import numpy as np
import torch

from . import ops
from ._lib import lib


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def pad_palette(palette):
    """The reference zero-pads its palette lists to 256 x 3 entries (mtmadise.py:97-103)."""
    pal = list(int(v) for v in palette)
    assert len(pal) <= 768 and all(0 <= v <= 255 for v in pal)
    return pal + [0] * (768 - len(pal))


_palette_cache = {}


def _device_palette(palette, device):
    key = (tuple(palette), device)
    t = _palette_cache.get(key)
    if t is None:
        t = torch.tensor(pad_palette(palette), dtype=torch.uint8, device=device)
        _palette_cache[key] = t
    return t


def convert_label_to_rgb(label, palette):
    """label: int64 [B, 1, H, W] (255 = ignore); palette: list of up to 768 ints.  Returns (color_label f32
    [B, 3, H, W] in [-1, 1], valid_mask f32 [B, 1, H, W]) like the reference -- without the D2H / PIL / H2D trip."""
    ops._need_cuda(label)
    B, C, H, W = label.shape
    assert C == 1
    lab = label.to(torch.int64).contiguous()
    pal = _device_palette(palette, label.device)
    rgb = torch.empty((B, 3, H, W), dtype=torch.float32, device=label.device)
    valid = torch.empty((B, 1, H, W), dtype=torch.float32, device=label.device)
    ops.check(lib.madm_label_to_rgb(lab.data_ptr(), pal.data_ptr(), rgb.data_ptr(), valid.data_ptr(), B, H * W, _s()),
              "madm_label_to_rgb")
    return rgb, valid


def pseudo_labels(ema_logits, size, pseudo_threshold):
    """ema_logits: f32 [B, K, h, w] (teacher head output); size: (H, W) of the input.  Returns (pseudo_prob f32
    [B, H, W], pseudo_label i64 [B, H, W], pseudo_weight f32 [B, H, W] = fraction of confident pixels, broadcast)
    with NO host synchronisation (the reference reads the fraction back with .item())."""
    ops._need_cuda(ema_logits)
    B, K, h, w = ema_logits.shape
    H, W = size
    x = ema_logits.float().contiguous()
    if (h, w) != (H, W):
        x = ops.resize_bilinear_nchw(x, H, W)   # F.interpolate(mode="bilinear", align_corners=False)
    prob = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
    label = torch.empty((B, H, W), dtype=torch.int64, device=x.device)
    count = torch.zeros(1, dtype=torch.int64, device=x.device)
    ops.check(lib.madm_pseudo_label(x.data_ptr(), prob.data_ptr(), label.data_ptr(), count.data_ptr(), B, K, H * W,
                                    float(pseudo_threshold), _s()), "madm_pseudo_label")
    # python float64 division in the reference (torch.sum(...).item() / ps_size), then a float32 tensor: the same rounding
    pseudo_val = (count.to(torch.float64) / float(B * H * W)).to(torch.float32)          # stays on the device
    pseudo_weight = pseudo_val * torch.ones(prob.shape, device=x.device)
    return prob, label, pseudo_weight


def label_classes(labels):
    """Sorted class values present in ``labels`` (== torch.unique(labels) for values 0..255): a 256-flag kernel and a
    1 KB read-back instead of a device sort."""
    ops._need_cuda(labels)
    lab = labels.to(torch.int64).contiguous()
    flags = torch.zeros(256, dtype=torch.int32, device=labels.device)
    ops.check(lib.madm_label_presence(lab.data_ptr(), lab.numel(), flags.data_ptr(), _s()), "madm_label_presence")
    return torch.nonzero(flags.cpu(), as_tuple=False).flatten().to(torch.int64)


def get_class_choices(labels, rng=np.random):
    """ClassMix class choice with the reference's RNG calls (dacs_transforms.py:81-90; like there the candidate
    classes are those of the WHOLE batch): one int64 tensor of chosen class values per image."""
    ops._need_cuda(labels)
    classes = label_classes(labels)
    chosen = []
    for _ in labels:
        nclasses = classes.shape[0]
        class_choice = rng.choice(nclasses, int((nclasses + nclasses % 2) / 2), replace=False)
        chosen.append(classes[torch.Tensor(class_choice).long()])
    return chosen


def get_class_masks(labels, rng=np.random):
    """dacs_transforms.get_class_masks: one f32 mask [1, 1, H, W] per image, on the device."""
    return [class_mix(label, chosen)[0].unsqueeze(0) for label, chosen in zip(labels, get_class_choices(labels, rng))]


def class_mix(label0, chosen, img0=None, img1=None, label1=None):
    """One image pair: mask = label0 in ``chosen`` (f32 [1, H, W]); mixed image / label = mask * x0 + (1 - mask) * x1
    (``one_mix``, dacs_transforms.py:100-111).  label0: i64 [1, H, W] or [H, W]; img*: f32 [C, H, W]."""
    ops._need_cuda(label0, img0, img1, label1)
    lab0 = label0.to(torch.int64).contiguous()
    H, W = lab0.shape[-2:]
    sel = torch.zeros(256, dtype=torch.uint8)
    sel[torch.as_tensor(chosen, dtype=torch.int64).cpu()] = 1
    sel = sel.to(lab0.device)
    mask = torch.empty((1, H, W), dtype=torch.float32, device=lab0.device)
    img_out = lab_out = None
    C = 0
    if img0 is not None:
        img0, img1 = img0.float().contiguous(), img1.float().contiguous()
        C = img0.shape[0]
        img_out = torch.empty_like(img0)
    if label1 is not None:
        label1 = label1.to(torch.int64).contiguous()
        lab_out = torch.empty_like(lab0)
    p = lambda t: None if t is None else t.data_ptr()
    ops.check(lib.madm_class_mix(lab0.data_ptr(), p(label1), sel.data_ptr(), p(img0), p(img1), C, H * W, mask.data_ptr(),
                                 p(img_out), p(lab_out), _s()), "madm_class_mix")
    return mask, img_out, lab_out
