"""One optimisation step of the LdmRocm extractor with a torch-side loss (the reference's step is
engine/train_loop.py:203-217 -- backward, unscale, clip_grad_norm_, optimizer.step -- plus CMDISE._update_ema,
modeling/meta_arch/cmdise.py:337-349):

    feats = ldm(batch)                 HIP forward (no-grad) + ONE autograd node over the UNet stage
    loss  = loss_fn(feats)             torch: whatever consumes the features (projections / head / criterion)
    loss.backward()                    torch autograd -> _UNetTapsFn.backward -> backward.unet_backward (HIP kernels);
                                       gradients accumulate straight into the flat fp32 gradient buffer
    all-reduce(mean)                   dist.GradBucketReducer over that buffer (multi-GPU only)
    clip + AdamW, EMA                  optim.FlatAdamW / optim.ema_update: three launches over the flat buffers

The trainable set is whatever has requires_grad on the UNet (``LdmRocm._freeze`` modes, or the active LoRA matrices)
plus ``extra_params`` (e.g. the prompt / time embeddings of BasePromptTimeGenerator, ldm_base.py:632-717).
"""
import torch

from . import optim
from .dist import GradBucketReducer


class ExtractorTrainer:
    def __init__(self, ldm, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, clip_grad=None, dist=None,
                 ema_alpha=None, extra_params=()):
        self.ldm = ldm
        params = [p for p in ldm.unet.parameters() if p.requires_grad] + [p for p in extra_params if p.requires_grad]
        assert params, "nothing to train: set requires_grad on the UNet / LoRA parameters first"
        self.flat = optim.FlatParams(params, with_grad=True)
        self.opt = optim.FlatAdamW(self.flat, lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.clip_grad = clip_grad
        self.reducer = GradBucketReducer(self.flat.grad, dist)
        self.ema_alpha = ema_alpha
        self.ema = self.flat.flat.clone() if ema_alpha is not None else None

    def step(self, batched_inputs, loss_fn, input_modal="rgb", **kwargs):
        """Returns (loss value as a python float, total gradient norm or None)."""
        self.flat.grad.zero_()
        feats = self.ldm(batched_inputs, input_modal, **kwargs)
        loss = loss_fn(feats)
        loss.backward()
        self.reducer.finish()
        norm = self.opt.step(clip_grad=self.clip_grad)
        if self.ema is not None:
            optim.ema_update(self.ema, self.flat.flat, self.ema_alpha)
        return float(loss.detach()), norm
