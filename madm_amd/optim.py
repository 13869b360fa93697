"""Flat-storage optimizer / EMA for the training-step tail.

``FlatParams`` re-points every parameter of a module list at a slice of ONE contiguous fp32 buffer (and likewise the
gradients), so the clip / AdamW / EMA passes are single launches over ~870 M elements instead of Python loops over
~700 tensors (engine/train_loop.py:203-217, config_files/common/optim.py:8-17, modeling/meta_arch/cmdise.py:337-349).
The forward-only modules of this package never allocate gradients; these classes are the device-side pieces a
training loop needs once backward kernels exist."""
import ctypes

import torch

from ._lib import lib, check
from .ops import _stream, _need_cuda


class FlatParams:
    """Parameters of ``modules`` (in ``parameters()`` order, de-duplicated) as views into one flat fp32 buffer."""

    def __init__(self, params, with_grad=True, align=4):
        """``align``: every tensor starts at a multiple of ``align`` elements (4 = 16 bytes; 1024 for the per-tensor
        hyper-parameter table of ``TableAdamW``)."""
        self.params = []
        seen = set()
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        assert all(p.dtype == torch.float32 for p in self.params)
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + align - 1) // align * align     # keep every tensor (at least) 16-byte aligned
        self.numel = n
        self.offsets, self.align = offs, align
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev) if with_grad else None
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                view = self.flat[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                if with_grad:
                    p.grad = self.grad[o:o + p.numel()].view_as(p)


def grad_sumsq(flat_grad):
    _need_cuda(flat_grad)
    out = torch.zeros(1, dtype=torch.float64, device=flat_grad.device)
    check(lib.madm_sumsq_f32(flat_grad.data_ptr(), flat_grad.numel(), out.data_ptr(), _stream()), "madm_sumsq_f32")
    return out


class FlatAdamW:
    """torch.optim.AdamW semantics on a FlatParams (one parameter group), with the GradScaler unscale and the
    clip_grad_norm_ coefficient folded into the update."""

    def __init__(self, flat, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.step_count = 0

    def step(self, clip_grad=None, loss_scale=1.0):
        """Returns the (unscaled) total gradient norm as a python float when clipping, else None."""
        g = self.flat.grad
        scale = 1.0 / loss_scale
        norm = None
        if clip_grad is not None:
            norm = float(grad_sumsq(g).item()) ** 0.5 * scale      # one host sync, as clip_grad_norm_'s .item() users do
            scale *= min(1.0, clip_grad / (norm + 1e-6))
        self.step_count += 1
        check(lib.madm_adamw_step(self.flat.flat.data_ptr(), g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                  self.flat.numel, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                  self.step_count, scale, _stream()), "madm_adamw_step")
        # the kernel wrote through raw pointers: move the parameters' version counters like an in-place torch op would,
        # so the packed device operands derived from them (nn._Packed caches) are re-derived on the next forward
        torch.autograd.graph.increment_version(self.flat.params)
        return norm


def ema_update(ema_flat, param_flat, alpha):
    _need_cuda(ema_flat, param_flat)
    assert ema_flat.numel() == param_flat.numel() and ema_flat.dtype == torch.float32
    check(lib.madm_ema_update(ema_flat.data_ptr(), param_flat.data_ptr(), ema_flat.numel(), float(alpha), _stream()),
          "madm_ema_update")


class EmaPairs:
    """CMDISE._update_ema (modeling/meta_arch/cmdise.py:337-349) for a list of (teacher, student) parameter pairs: runs of
    pairs that are adjacent in memory on both sides (parameters living in FlatParams buffers, laid out in the same order)
    are updated by ONE madm_ema_update launch each; isolated tensors get their own launch.  Parameter versions are moved
    afterwards so packed operands derived from the teacher are refreshed."""

    def __init__(self, teacher_params, student_params):
        self.pairs = [(t, s) for t, s in zip(teacher_params, student_params)]
        assert all(t.shape == s.shape and t.dtype == torch.float32 and s.dtype == torch.float32 for t, s in self.pairs)
        self._spans = None
        self._key = None

    def _build(self):
        spans = []   # [teacher ptr, student ptr, bytes]
        for t, s in self.pairs:
            tp, sp, nb = t.data_ptr(), s.data_ptr(), t.numel() * 4
            if spans:
                lt, ls, lb = spans[-1]
                gap = tp - (lt + lb)
                if 0 <= gap < 4096 and sp - (ls + lb) == gap and (lb + gap) % 4 == 0:   # same (zero) padding on both sides
                    spans[-1][2] = lb + gap + nb
                    continue
            spans.append([tp, sp, nb])
        return spans

    def update(self, alpha):
        key = tuple((t.data_ptr(), s.data_ptr()) for t, s in self.pairs)
        if key != self._key:
            self._key, self._spans = key, self._build()
        for tp, sp, nb in self._spans:
            check(lib.madm_ema_update(ctypes.c_void_p(tp), ctypes.c_void_p(sp), nb // 4, float(alpha), _stream()),
                  "madm_ema_update")
        torch.autograd.graph.increment_version([t for t, _ in self.pairs])


def default_optimizer_params(model, lr, weight_decay, weight_decay_norm=0.0, weight_decay_bias=0.0, unet_lr=None):
    """get_default_optimizer_params_unet (utils/parameter_count.py:120-215) as used by config_files/common/optim.py:8-17:
    one (parameter, lr, weight_decay) triple per trainable parameter in ``named_modules`` order, de-duplicated; parameters
    of normalisation modules get ``weight_decay_norm``, parameters NAMED 'bias' get ``weight_decay_bias`` (the override
    wins, as there), modules whose name contains 'unet' get ``unet_lr``."""
    from . import nn as mnn
    from . import head as mhead
    norm_types = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.SyncBatchNorm, torch.nn.GroupNorm,
                  torch.nn.InstanceNorm1d, torch.nn.InstanceNorm2d, torch.nn.InstanceNorm3d, torch.nn.LayerNorm,
                  torch.nn.LocalResponseNorm, mnn.GroupNorm, mnn.LayerNorm, mhead._BN)
    out, memo = [], set()
    for module_name, module in model.named_modules():
        for pname, p in module.named_parameters(recurse=False):
            if not p.requires_grad or id(p) in memo:
                continue
            memo.add(id(p))
            wd, plr = weight_decay, lr
            if isinstance(module, norm_types) and weight_decay_norm is not None:
                wd = weight_decay_norm
            if 'unet' in module_name and unet_lr is not None:
                plr = unet_lr
            if pname == "bias" and weight_decay_bias is not None:
                wd = weight_decay_bias
            out.append((p, plr, wd))
    return out


class TableAdamW:
    """torch.optim.AdamW over per-parameter groups (``default_optimizer_params``) as ONE launch on flat storage: every
    tensor carries its own lr / weight decay / step count; tensors that received no gradient this step are skipped like
    torch skips ``grad is None`` parameters.  The GradScaler unscale and the clip coefficient ride on the gradient scale."""

    def __init__(self, param_table, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p, _, _ in param_table]
        self.base_lr = [float(l) for _, l, _ in param_table]
        self.wd = [float(w) for _, _, w in param_table]
        self.flat = FlatParams(self.params, with_grad=True, align=1024)
        self.betas, self.eps = betas, eps
        self.m = torch.zeros_like(self.flat.flat)
        self.v = torch.zeros_like(self.flat.flat)
        import numpy as np
        self.steps = np.zeros(len(self.params), dtype=np.int64)
        self._base_lr_np, self._wd_np = np.asarray(self.base_lr, dtype=np.float64), np.asarray(self.wd, dtype=np.float64)
        dev = self.flat.flat.device
        ct = torch.empty(self.flat.numel // 1024, dtype=torch.int32)
        for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets)):
            ct[o // 1024:(o + (p.numel() + 1023) // 1024 * 1024) // 1024] = i
        self.chunk_tensor = ct.to(dev)
        self._hyper_host = torch.zeros((len(self.params), 4), dtype=torch.float32).pin_memory() if dev.type == "cuda" \
            else torch.zeros((len(self.params), 4), dtype=torch.float32)
        self._hyper = torch.zeros((len(self.params), 4), dtype=torch.float32, device=dev)
        self.lr_factor = 1.0           # the LR scheduler's multiplier (WarmupParamScheduler in the reference's config)

    def zero_grad(self):
        self.flat.grad.zero_()

    def step(self, clip_grad=None, loss_scale=1.0, touched=None, on_synced=None):
        """``touched``: ids of the parameters that received a gradient (None: all).  Returns (total gradient norm or None,
        stepped: False when the norm is not finite -- GradScaler's inf / nan skip).  ``on_synced``: called right after the
        step's one host sync and BEFORE any state is touched (MadmTrainer's deferred input-range assert: an exception there
        leaves parameters, moments and step counts as they were)."""
        g = self.flat.grad
        scale = 1.0 / loss_scale
        norm = float(grad_sumsq(g).item()) ** 0.5 * scale       # one host sync (GradScaler.step syncs on found_inf too)
        if on_synced is not None:
            on_synced()
        if not (norm == norm and norm != float("inf")):
            return norm, False
        if clip_grad is not None:
            scale *= min(1.0, clip_grad / (norm + 1e-6))
        h = self._hyper_host
        b1, b2 = self.betas
        # the per-tensor table in ONE vectorised numpy statement per column: as a Python loop of tensor element assignments this was
        # 3 077 aten::copy_ calls = ~9 ms of host time per step, exposed behind the step's host sync (round 6, tools/exp/train_aten_sites.py)
        import numpy as np
        if touched is None:
            mask = np.ones(len(self.params), dtype=bool)
        else:
            mask = np.fromiter((id(p) in touched for p in self.params), dtype=bool, count=len(self.params))
        steps = np.asarray(self.steps, dtype=np.int64)
        steps[mask] += 1
        self.steps = steps
        t = steps.astype(np.float64)
        hn = h.numpy()                      # (pinned host tensor: shares memory)
        hn[mask, 0] = (self._base_lr_np[mask] * self.lr_factor).astype(np.float32)
        hn[mask, 1] = self._wd_np[mask].astype(np.float32)
        hn[:, 2] = np.where(mask, 1.0 - np.power(b1, t), 0.0).astype(np.float32)
        hn[mask, 3] = np.sqrt(1.0 - np.power(b2, t[mask])).astype(np.float32)
        self._hyper.copy_(h, non_blocking=True)
        check(lib.madm_adamw_step_table(self.flat.flat.data_ptr(), g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                        self.flat.numel, self.chunk_tensor.data_ptr(), self._hyper.data_ptr(), b1, b2, self.eps,
                                        scale, _stream()), "madm_adamw_step_table")
        torch.autograd.graph.increment_version(self.flat.params)
        return norm, True
