"""Backward passes composed from the C-ABI gradient kernels (first slice).

The reference trains through torch autograd (``losses.backward()``, engine/train_loop.py:203-217); here the gradient of
a module is an explicit function over the same channels-last ``Tok`` tensors its forward uses, built from
``madm_conv2d_wgrad`` / the forward conv on repacked weights (data gradient) / ``madm_groupnorm_bwd_*`` /
``madm_layernorm_bwd``.  Activations are RECOMPUTED from the block input (the forward keeps nothing but what its caller
holds), which is also how the 288 GB budget would be spent at bs = 2: no activation stash at all.

Parameter gradients come back as ``{parameter name: f32 tensor in the nn.Parameter's own shape}`` -- what a
``FlatParams`` gradient buffer (madm_amd/optim.py) takes.  Built so far: Conv2d (stride 1 / stride 2 with both paddings /
nearest-2x upsample), Linear, GroupNorm(+act), LayerNorm and diffusers' ResnetBlock2D (ldm_diffusers.py:290,333,387,435
call sites); the transformer blocks (attention backward) and the autograd wiring of the whole UNet are not.
"""
import torch

from . import ops, packing
from .nn import Tok


def _colsum_per_image(t, B, HW):
    """f32 [B, C] per-image column sums of a [B*HW, C] gradient (bias / time-row gradients): the channel-sum kernel of
    the GroupNorm statistics."""
    sums = ops.new_chsums(B, t.shape[1], t.device)
    ops.groupnorm_stats(t, B, HW, sums)
    return sums[:, :, 0].float()


def conv2d_backward(conv, x, dout, x2=None, need_dx=True, dres=None, upsample=False):
    """Gradients of ``conv(x, x2, upsample=...)`` (plain conv: the caller handles a fused norm) for the ``nn.Conv2d``
    twin ``conv``: returns (dx_cat [M_in, C1(+C2)] or None, {"weight": ..., "bias": ...}).  ``dres`` (stride-1 layers)
    is added to dx.  Stride-2 layers (Downsample2D, both paddings) run the stride-1 data gradient on the zero-inserted
    ``dout``; the nearest-2x upsample conv (Upsample2D) runs it at the upsampled size and sum-pools 2 x 2."""
    dtype = x.t.dtype
    kt = ops.k_tile(dtype)
    k = conv.kernel_size
    pad = 0 if conv.asym_pad else conv.padding
    OH, OW = conv.out_hw(x.H, x.W, upsample)
    splits = None if x2 is None else [x.C, x2.C]
    dwp = ops.conv2d_wgrad(x.t, dout, x.B, x.H, x.W, x2=None if x2 is None else x2.t, KH=k, KW=k, stride=conv.stride,
                           pad_t=pad, pad_l=pad, OH=OH, OW=OW, upsample=upsample)
    dw = packing.unpack_conv_weight_grad(dwp[:conv.out_channels], conv.in_channels, k, k, kt, splits=splits)
    grads = {"weight": dw}
    if conv.bias is not None:
        grads["bias"] = _colsum_per_image(dout, x.B, OH * OW).sum(0)[:conv.out_channels]
    dx = None
    if need_dx:
        wp, _ = conv.packed(dtype, splits)
        wt = ops.pack_dgrad_weights(wp, k * k)
        if conv.stride == 1 and not upsample:
            dx = ops.conv2d_dgrad(dout, wt, x.B, x.H, x.W, C=wt.shape[0], KH=k, KW=k, pad_t=pad, pad_l=pad, residual=dres)
        elif conv.stride == 2 and not upsample:
            assert dres is None
            dz = ops.zero_insert2x(dout, x.B, OH, OW, x.H, x.W)
            dx = ops.conv2d_dgrad(dz, wt, x.B, x.H, x.W, C=wt.shape[0], KH=k, KW=k, pad_t=pad, pad_l=pad)
        elif conv.stride == 1 and upsample:
            assert dres is None
            du = ops.conv2d_dgrad(dout, wt, x.B, OH, OW, C=wt.shape[0], KH=k, KW=k, pad_t=pad, pad_l=pad)
            dx = ops.sumpool2x2(du, x.B, x.H, x.W)
        else:
            raise NotImplementedError(f"data gradient of stride {conv.stride} upsample {upsample}")
    return dx, grads


def linear_backward(lin, x, dout, need_dx=True):
    """Gradients of ``lin(x)``: (dx [M, K] or None, {"weight", "bias"})."""
    dtype = x.dtype
    M = x.shape[0]
    dwp = ops.conv2d_wgrad(x, dout, 1, M, 1)
    grads = {"weight": dwp[:, :lin.in_features].contiguous()}
    if lin.bias is not None:
        grads["bias"] = _colsum_per_image(dout, 1, M)[0]
    dx = None
    if need_dx:
        wp, _ = lin.packed(dtype)
        dx = ops.conv2d_dgrad(dout, ops.pack_dgrad_weights(wp, 1), 1, M, 1, C=wp.shape[1])
    return dx, grads


def resnet_block_backward(block, x, dout, temb_row=None, skip=None):
    """Backward of ``sd_unet.ResnetBlock2D.forward(x, temb_row, skip)`` for the output gradient ``dout`` [M, Cout]:

        h   = conv1(silu(norm1([x | skip]))) + b1 + temb_row[image]
        out = conv2(silu(norm2(h))) + b2 + shortcut([x | skip])

    returns (dx, dskip or None, dtemb_row f32 [B, Cout] or None, {"norm1.weight": ..., "conv1.weight": ..., ...}).
    The intermediate h and the two activated tensors are recomputed here."""
    B, H, W, HW = x.B, x.H, x.W, x.HW
    srcs = [x] if skip is None else [x, skip]
    for s in srcs:
        if s.stats is None:
            s.stats = ops.new_chsums(B, s.C, s.t.device)
            ops.groupnorm_stats(s.t, B, HW, s.stats)
    n1, n2 = block.norm1, block.norm2
    g1, b1 = n1.weight.detach().float(), n1.bias.detach().float()
    g2, b2 = n2.weight.detach().float(), n2.bias.detach().float()
    grads = {}

    # ---- recompute: a1 = silu(norm1(.)), h (raw conv1 output incl. bias and time row, with its channel sums) ----
    a1 = n1(x, silu=True, x2=skip)
    h = block.conv1(a1, rowvec=temb_row)
    a2 = n2(h, silu=True)

    # ---- conv2 and norm2 ----
    da2, g = conv2d_backward(block.conv2, a2, dout)
    grads.update({"conv2." + k_: v for k_, v in g.items()})
    (dh,), dg2, db2 = ops.groupnorm_backward([h.t], da2, B, HW, n2.num_groups, g2, b2, n2.eps, [h.stats], act="silu")
    grads["norm2.weight"], grads["norm2.bias"] = dg2, db2

    # ---- time row and conv1 ----
    dtemb_row = None
    if temb_row is not None:
        dtemb_row = _colsum_per_image(dh, B, HW)[:, :block.conv1.out_channels].contiguous()
    da1, g = conv2d_backward(block.conv1, a1, dh)
    grads.update({"conv1." + k_: v for k_, v in g.items()})

    # ---- shortcut: its data gradient joins norm1's dx as ``dres`` ----
    if block.conv_shortcut is None:
        dsc = dout
    else:
        dsc, g = conv2d_backward(block.conv_shortcut, x, dout, x2=skip)
        grads.update({"conv_shortcut." + k_: v for k_, v in g.items()})
    dxs, dg1, db1 = ops.groupnorm_backward([s.t for s in srcs], da1, B, HW, n1.num_groups, g1, b1, n1.eps,
                                           [s.stats for s in srcs], act="silu", dres=dsc)
    grads["norm1.weight"], grads["norm1.bias"] = dg1, db1
    dx = Tok(dxs[0], B, H, W)
    dskip = Tok(dxs[1], B, H, W) if skip is not None else None
    return dx, dskip, dtemb_row, grads
