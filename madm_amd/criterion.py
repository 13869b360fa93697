"""``CmdiseCriterion`` on the HIP kernels (/root/reference/modeling/criterion.py:110-254): the losses of the
self-training step as explicit forward + gradient functions over channels-last logit tokens.

The reference computes ``F.interpolate(pred, label size) -> F.cross_entropy(reduction='none', ignore_index=255) *
pixel_weight -> .mean()`` and, per ``vae_decoder_loss`` entry, ``sum(|pred - gt| * nearest(mask)) / numel * weight`` in
torch and leaves the gradients to autograd; here ``*_forward`` returns the loss scalar (an f32 device tensor, no host
sync) plus what ``*_backward`` needs, and ``*_backward`` returns the gradient for a given upstream scalar ``g`` (a device
tensor: the GradScaler scale rides on it).  Only the shipped reduction ('mean', no class weights) is built."""
import torch

from . import ops
from .nn import Tok


class CmdiseCriterion(torch.nn.Module):
    def __init__(self, num_classes=19, pseudo_threshold=0.968, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        if reduction != 'mean' or class_weight is not None:
            raise NotImplementedError("CmdiseCriterion: only reduction='mean' without class weights (every shipped config)")
        self.num_classes, self.pseudo_threshold, self.reduction = num_classes, pseudo_threshold, reduction
        self.class_weight, self.loss_weight = class_weight, loss_weight

    # ---- cross entropy on logit tokens --------------------------------------------------------------------------
    def ce_forward(self, logits, K, label, pixel_weight=None, ignore_index=255):
        """logits: Tok of f32 tokens [B*h*w, >=K]; label: i64 [B, H, W]; pixel_weight: f32 [B, H, W] or None.
        Returns (loss scalar f32 [], ctx)."""
        B, H, W = label.shape
        up = logits
        if (logits.H, logits.W) != (H, W):     # F.interpolate(..., mode='bilinear', align_corners=False) (:172,:183)
            up = Tok(ops.resize_bilinear(logits.t, B, logits.H, logits.W, H, W), B, H, W)
        M = B * H * W
        label = label.contiguous()
        pw = None if pixel_weight is None else pixel_weight.float().contiguous()
        s = torch.zeros(1, dtype=torch.float64, device=label.device)
        ops.softmax_ce(up.t, K, label, pw, ignore_index, loss_sum=s)
        loss = (s * (self.loss_weight / M)).to(torch.float32).reshape(())
        return loss, dict(logits=logits, up=up, K=K, label=label, pw=pw, ignore=ignore_index, M=M)

    def ce_backward(self, ctx, g, grad_dtype):
        """d loss / d logits as tokens [B*h*w, k_tile] of ``grad_dtype`` for the upstream scalar gradient g (device f32)."""
        logits, up = ctx["logits"], ctx["up"]
        _, d = ops.softmax_ce(up.t, ctx["K"], ctx["label"], ctx["pw"], ctx["ignore"], gscale=g.reshape(1).float().contiguous(),
                              coef=self.loss_weight / ctx["M"], grad_dtype=grad_dtype)
        if up is not logits:
            d = ops.resize_bilinear_backward(d, up.B, logits.H, logits.W, up.H, up.W)
        return d

    # ---- masked L1 / L2 on the latents (vae_decoder_loss entries, :236-246) --------------------------------------
    @staticmethod
    def decoder_loss_forward(pred, gt, mask, loss_weight, loss_type):
        """pred / gt: f32 [B, 4, h, w]; mask: f32 [B, 1, H, W].  Returns (loss scalar, ctx)."""
        pred, gt, mask = pred.float().contiguous(), gt.float().contiguous(), mask.float().contiguous()
        s = torch.zeros(1, dtype=torch.float64, device=pred.device)
        ops.masked_l1(pred, gt, mask, l2=(loss_type != 'L1'), loss_sum=s)
        coef = float(loss_weight) / pred.numel()
        return (s * coef).to(torch.float32).reshape(()), dict(pred=pred, gt=gt, mask=mask, l2=(loss_type != 'L1'), coef=coef)

    @staticmethod
    def decoder_loss_backward(ctx, g):
        _, d = ops.masked_l1(ctx["pred"], ctx["gt"], ctx["mask"], l2=ctx["l2"], gscale=g.reshape(1).float().contiguous(),
                             coef=ctx["coef"], want_grad=True)
        return d
