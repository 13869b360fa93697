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


# ----------------------------------------------------------------------------- transformer block
def _layer_names(layer, prefix):
    """parameter-name prefixes of one projection layer: plain Linear or peft-shaped LoraLinear."""
    from .sd_unet import LoraLinear
    return (prefix + ".base_layer") if isinstance(layer, LoraLinear) else prefix


def fused_proj_backward(fp, names, x, dout, need_dx=True, base_grads=True):
    """Backward of ``sd_unet._FusedProj.forward(x)`` (several Linear layers sharing one input as ONE GEMM, LoRA as a
    K-extension): returns (dx [M, Kpad] or None, grads).  ``names`` are the attribute names of the fused layers (for the
    gradient keys); LoRA A / B gradients always come back, the frozen base weights / biases only with ``base_grads``.

        t = x Abar^T,  out = [x | t] [W | Bx]^T + b     (Abar = stacked scaling * A, Bx = block-placed B)
        d[x | t] = dout [W | Bx];  dBx = dout^T t;  dAbar = dt^T x;  dx = dx_part + dt Abar;  dW = dout^T x
    """
    from .sd_unet import LoraLinear, LORA_PAD, _base
    dtype = x.dtype
    M = x.shape[0]
    Wp, bias, A = fp._cache_get((dtype,), lambda: fp._build(dtype))
    layers = fp._layers
    bases = [_base(l) for l in layers]
    K = bases[0].in_features
    n_off = [0]
    for b_ in bases:
        n_off.append(n_off[-1] + b_.out_features)
    grads = {}
    if base_grads:
        dW = ops.conv2d_wgrad(x, dout, 1, M, 1)
        colsum = _colsum_per_image(dout, 1, M)[0] if bias is not None else None
        for i, (l, nm) in enumerate(zip(layers, names)):
            pre = _layer_names(l, nm)
            grads[pre + ".weight"] = dW[n_off[i]:n_off[i + 1], :K].contiguous()
            if bases[i].bias is not None:
                grads[pre + ".bias"] = colsum[n_off[i]:n_off[i + 1]].contiguous()
    if A is None:
        dx = ops.conv2d_dgrad(dout, ops.pack_dgrad_weights(Wp, 1), 1, M, 1, C=Wp.shape[1]) if need_dx else None
        return dx, grads
    Kp = Wp.shape[1] - LORA_PAD
    t = ops.linear(x, A)                                                  # recomputed [M, LORA_PAD]
    dcat = ops.conv2d_dgrad(dout, ops.pack_dgrad_weights(Wp, 1), 1, M, 1, C=Wp.shape[1])
    dxb, dt = dcat[:, :Kp], dcat[:, Kp:]
    dBx = ops.conv2d_wgrad(t, dout, 1, M, 1)                              # [Ntot, LORA_PAD]
    dAbar = ops.conv2d_wgrad(x, dt, 1, M, 1)                              # [LORA_PAD, Kpad]
    r0 = 0
    for i, (l, nm) in enumerate(zip(layers, names)):
        if not isinstance(l, LoraLinear):
            continue
        for n in l.active():
            r = l.lora_A[n].weight.shape[0]
            grads[f"{nm}.lora_A.{n}.weight"] = (dAbar[r0:r0 + r, :K] * l.scaling[n]).contiguous()
            grads[f"{nm}.lora_B.{n}.weight"] = dBx[n_off[i]:n_off[i + 1], r0:r0 + r].contiguous()
            r0 += r
    dx = None
    if need_dx:   # dx = dx_part + dt Abar: the first addend rides on the second GEMM's residual input
        dx = ops.conv2d_dgrad(dt, ops.pack_dgrad_weights(A, 1), 1, M, 1, C=A.shape[1], residual=dxb)
    return dx, grads


def attention_module_backward(attn, x, dout, B, L, ctx=None, Lk=None, kv=None, base_grads=True):
    """Backward of ``sd_unet.Attention.forward(x, B, L, ctx, Lk, residual, kv)`` w.r.t. x, the K/V source and the
    parameters (the residual's gradient is ``dout`` itself and is left to the caller).  Returns
    (dx [M, C], dkv, grads): dkv is the gradient of the precomputed ``kv`` view when one was given (the UNet-wide
    batched K/V projection), else the gradient of ``ctx`` (cross) or None (self).  q / k / v / o are recomputed."""
    C = attn.heads * attn.dim_head
    H, D = attn.heads, attn.dim_head
    grads = {}
    if not attn.is_cross:
        f_qkv = attn._fused("_f_qkv", ("to_q", "to_k", "to_v"))
        qkv = f_qkv(x)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        Lk = L
    else:
        f_q = attn._fused("_f_q", ("to_q",))
        q = f_q(x)
        kvt = kv if kv is not None else attn._fused("_f_kv", ("to_k", "to_v"))(ctx)
        k, v = kvt[:, :C], kvt[:, C:]
    o = ops.attention(q, k, v, B, H, L, Lk, D, attn.scale)
    f_out = attn._fused("_f_out", ("to_out",))
    d_o, g = fused_proj_backward(f_out, ("to_out.0",), o, dout, base_grads=base_grads)
    grads.update(g)
    d_o = d_o[:, :C] if d_o.shape[1] != C else d_o
    if not attn.is_cross:
        dqkv = torch.empty_like(qkv)
        ops.attention_backward(q, k, v, o, d_o, B, H, L, Lk, D, attn.scale,
                               outs=(dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:]))
        dx, g = fused_proj_backward(f_qkv, ("to_q", "to_k", "to_v"), x, dqkv, base_grads=base_grads)
        grads.update(g)
        return dx, None, grads
    dq = torch.empty((B * L, C), dtype=x.dtype, device=x.device)
    dkv = torch.empty((B * Lk, 2 * C), dtype=x.dtype, device=x.device)
    ops.attention_backward(q, k, v, o, d_o, B, H, L, Lk, D, attn.scale, outs=(dq, dkv[:, :C], dkv[:, C:]))
    dx, g = fused_proj_backward(f_q, ("to_q",), x, dq, base_grads=base_grads)
    grads.update(g)
    if kv is not None:
        return dx, dkv, grads
    dctx, g = fused_proj_backward(attn._fused("_f_kv", ("to_k", "to_v")), ("to_k", "to_v"), ctx, dkv,
                                  base_grads=base_grads)
    grads.update(g)
    return dx, dctx, grads


def feed_forward_backward(ff, x, dout, base_grads=True):
    """Backward of ``sd_unet.FeedForward.forward(x)`` = Linear(GEGLU(x)) (the residual is the caller's): (dx, grads).
    The GEGLU pre-activations are recomputed by the same GEMM with a plain epilogue."""
    geglu, lin = ff.net[0], ff.net[2]
    dtype = x.dtype
    M = x.shape[0]
    g_out = geglu(x)                                                      # [M, 4C]
    dg, g2 = linear_backward(lin, g_out, dout)
    grads = {"net.2." + k_: v for k_, v in g2.items()} if base_grads else {}
    dg = dg[:, :g_out.shape[1]] if dg.shape[1] != g_out.shape[1] else dg
    w, b = geglu._cache_get((dtype,), lambda: packing.pack_geglu_weight(
        geglu.proj.weight.detach().float(), geglu.proj.bias.detach().float(), dtype, ops.k_tile(dtype)))
    pre = ops.linear(x, w, bias=b)                                        # [M, 8C] interleaved (value, gate)
    dpre = ops.geglu_backward(pre, dg.contiguous())
    if base_grads:
        dwi = ops.conv2d_wgrad(x, dpre, 1, M, 1)[:, :geglu.proj.in_features]
        dbi = _colsum_per_image(dpre, 1, M)[0]
        half = dwi.shape[0] // 2
        # rows are interleaved (value_j, gate_j): back to diffusers' (value | gate) chunk order
        grads["net.0.proj.weight"] = torch.cat([dwi[0::2], dwi[1::2]], 0).contiguous()
        grads["net.0.proj.bias"] = torch.cat([dbi[0::2], dbi[1::2]], 0).contiguous()
        assert grads["net.0.proj.weight"].shape[0] == 2 * half
    dx = ops.conv2d_dgrad(dpre, ops.pack_dgrad_weights(w, 1), 1, M, 1, C=w.shape[1])
    return dx, grads


def transformer_block_backward(blk, h, dout, B, L, ctx, Lk, kv=None, base_grads=True):
    """Backward of ``sd_unet.BasicTransformerBlock.forward(h, B, L, ctx, Lk)`` (ctx: the [B*Lk, 768] prompt tokens, or
    ``kv`` = this block's precomputed cross-attention K/V view): returns (dh, dctx or dkv, grads).

        h1 = h + attn1(LN1(h));  h2 = h1 + attn2(LN2(h1), ctx);  out = h2 + ff(LN3(h2))

    Every intermediate is recomputed from h; the residual gradients ride on the LayerNorm backward kernels' ``dres``."""
    grads = {}

    def ln(norm, t):
        return ops.layernorm(t, norm.weight.detach(), norm.bias.detach(), norm.eps)

    n1 = ln(blk.norm1, h)
    h1 = blk.attn1(n1, B, L, residual=h)
    n2 = ln(blk.norm2, h1)
    h2 = blk.attn2(n2, B, L, ctx=ctx, Lk=Lk, residual=h1, kv=kv)
    n3 = ln(blk.norm3, h2)

    dn3, g = feed_forward_backward(blk.ff, n3, dout, base_grads=base_grads)
    grads.update({"ff." + k_: v for k_, v in g.items()})
    C = h.shape[1]
    dh2, dg, db = ops.layernorm_backward(h2, _cols(dn3, C), blk.norm3.weight.detach(), blk.norm3.eps, dres=dout)
    grads["norm3.weight"], grads["norm3.bias"] = dg, db

    dn2, dkv, g = attention_module_backward(blk.attn2, n2, dh2, B, L, ctx=ctx, Lk=Lk, kv=kv, base_grads=base_grads)
    grads.update({"attn2." + k_: v for k_, v in g.items()})
    dh1, dg, db = ops.layernorm_backward(h1, _cols(dn2, C), blk.norm2.weight.detach(), blk.norm2.eps, dres=dh2)
    grads["norm2.weight"], grads["norm2.bias"] = dg, db

    dn1, _, g = attention_module_backward(blk.attn1, n1, dh1, B, L, base_grads=base_grads)
    grads.update({"attn1." + k_: v for k_, v in g.items()})
    dh, dg, db = ops.layernorm_backward(h, _cols(dn1, C), blk.norm1.weight.detach(), blk.norm1.eps, dres=dh1)
    grads["norm1.weight"], grads["norm1.bias"] = dg, db
    return dh, dkv, grads


def _cols(t, C):
    """dense [M, C] from a data gradient whose row was padded to the K-tile."""
    return t if t.shape[1] == C else t[:, :C].contiguous()
