"""Backward passes composed from the C-ABI gradient kernels.

The reference trains through torch autograd (``losses.backward()``, engine/train_loop.py:203-217); here the gradient of
a module is an explicit function over the same channels-last ``Tok`` tensors its forward uses, built from
``madm_conv2d_wgrad`` / the forward conv on repacked weights (data gradient) / ``madm_attention_bwd`` /
``madm_groupnorm_bwd_*`` / ``madm_layernorm_bwd`` / ``madm_geglu_bwd``.  Activations are RECOMPUTED from the block input
(the forward keeps nothing but what its caller holds): ``unet_backward`` re-runs the forward keeping only block inputs
and every block's backward recomputes its interior -- no activation stash beyond ~40 block-boundary tensors.

Parameter gradients come back as ``{parameter name: f32 tensor in the nn.Parameter's own shape}``; ``ldm_rocm._UNetTapsFn``
hands them to torch autograd (``p.grad``), ``train.ExtractorTrainer`` accumulates them into an ``optim.FlatParams`` buffer.
Levels, each tested against torch autograd (tests/test_ops_gpu.py, tests/test_parity_gpu.py):
  conv2d_backward (stride 1 / stride 2 with both paddings / nearest-2x upsample), linear_backward,
  resnet_block_backward (diffusers ResnetBlock2D; ldm_diffusers.py:290,333,387,435 call sites),
  fused_proj_backward (fused QKV GEMM with the peft LoRA K-extension, mtmadise.py:115-147), attention_module_backward,
  feed_forward_backward (GEGLU), transformer_block_backward, transformer2d_backward,
  unet_backward (ldm_diffusers.py:454-616 with 'after' taps :442-445).
Not here: the backward of the HIP projections / DAFormer head, and of the VAE -- its parameters are frozen
(LdmDiffusers._freeze) and its input images do not require grad, so the reference's autograd never enters it either.
"""
from types import SimpleNamespace

import torch

from . import ops, packing
from .nn import Tok


def _colsum_per_image(t, B, HW):
    """f32 [B, C] per-image column sums of a [B*HW, C] gradient (bias / time-row gradients)."""
    return ops.colsum(t, B, HW)


def conv2d_backward(conv, x, dout, x2=None, need_dx=True, dres=None, upsample=False, need_dw=True):
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
    if dout.shape[1] % kt:   # conv_out (4 channels): the data gradient is a forward conv over dout's channels, whose
        dout = torch.nn.functional.pad(dout, (0, packing.round_up(dout.shape[1], kt) - dout.shape[1]))   # count fills a K-tile
    grads = {}
    if need_dw:   # False: frozen layer (the reference's LoRA mode), only the data gradient flows
        db = ops.zeros_f32((dout.shape[1],), dout.device) if conv.bias is not None else None   # gathered by the same launch
        dwp = ops.conv2d_wgrad(x.t, dout, x.B, x.H, x.W, x2=None if x2 is None else x2.t, KH=k, KW=k,
                               stride=conv.stride, pad_t=pad, pad_l=pad, OH=OH, OW=OW, upsample=upsample, dbias=db)
        grads["weight"] = packing.unpack_conv_weight_grad(dwp[:conv.out_channels], conv.in_channels, k, k, kt,
                                                          splits=splits)
        if conv.bias is not None:
            grads["bias"] = db[:conv.out_channels]
    dx = None
    if need_dx:
        def build_wt():
            wp, _ = conv.packed(dtype, splits)
            if wp.shape[0] != dout.shape[1]:   # conv_out: dout was widened to a K-tile above
                wp = torch.nn.functional.pad(wp, (0, 0, 0, dout.shape[1] - wp.shape[0]))
            return ops.pack_dgrad_weights(wp, k * k)

        # transposed / tap-reversed weights live in the module's packed-operand cache (re-derived when a parameter's
        # version moves, like the forward operands)
        wt = conv._cache_get((dtype, tuple(splits) if splits else None, "dgrad"), build_wt)
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


def linear_backward(lin, x, dout, need_dx=True, need_dw=True):
    """Gradients of ``lin(x)``: (dx [M, K] or None, {"weight", "bias"})."""
    dtype = x.dtype
    M = x.shape[0]
    grads = {}
    if need_dw:
        db = ops.zeros_f32((dout.shape[1],), dout.device) if lin.bias is not None else None
        dwp = ops.conv2d_wgrad(x, dout, 1, M, 1, dbias=db)
        grads["weight"] = dwp[:, :lin.in_features].contiguous()
        if lin.bias is not None:
            grads["bias"] = db
    dx = None
    if need_dx:
        wt = lin._cache_get((dtype, "dgrad"), lambda: ops.pack_dgrad_weights(lin.packed(dtype)[0], 1))
        dx = ops.conv2d_dgrad(dout, wt, 1, M, 1, C=wt.shape[0])
    return dx, grads


def resnet_block_backward(block, x, dout, temb_row=None, skip=None, base_grads=True):
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
    da2, g = conv2d_backward(block.conv2, a2, dout, need_dw=base_grads)
    grads.update({"conv2." + k_: v for k_, v in g.items()})
    (dh,), dg2, db2 = ops.groupnorm_backward([h.t], da2, B, HW, n2.num_groups, g2, b2, n2.eps, [h.stats], act="silu")
    grads["norm2.weight"], grads["norm2.bias"] = dg2, db2

    # ---- time row and conv1 ----
    dtemb_row = None
    if temb_row is not None:
        dtemb_row = _colsum_per_image(dh, B, HW)[:, :block.conv1.out_channels].contiguous()
    da1, g = conv2d_backward(block.conv1, a1, dh, need_dw=base_grads)
    grads.update({"conv1." + k_: v for k_, v in g.items()})

    # ---- shortcut: its data gradient joins norm1's dx as ``dres`` ----
    if block.conv_shortcut is None:
        dsc = dout
    else:
        dsc, g = conv2d_backward(block.conv_shortcut, x, dout, x2=skip, need_dw=base_grads)
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
        colsum = ops.zeros_f32((dout.shape[1],), dout.device) if bias is not None else None
        dW = ops.conv2d_wgrad(x, dout, 1, M, 1, dbias=colsum)
        for i, (l, nm) in enumerate(zip(layers, names)):
            pre = _layer_names(l, nm)
            grads[pre + ".weight"] = dW[n_off[i]:n_off[i + 1], :K].contiguous()
            if bases[i].bias is not None:
                grads[pre + ".bias"] = colsum[n_off[i]:n_off[i + 1]].contiguous()
    Wt, At = fp._cache_get((dtype, "dgrad"), lambda: (ops.pack_dgrad_weights(Wp, 1),
                                                      None if A is None else ops.pack_dgrad_weights(A, 1)))
    if A is None:
        dx = ops.conv2d_dgrad(dout, Wt, 1, M, 1, C=Wp.shape[1]) if need_dx else None
        return dx, grads
    Kp = Wp.shape[1] - LORA_PAD
    t = ops.linear(x, A)                                                  # recomputed [M, LORA_PAD]
    dcat = ops.conv2d_dgrad(dout, Wt, 1, M, 1, C=Wp.shape[1])
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
        dx = ops.conv2d_dgrad(dt, At, 1, M, 1, C=A.shape[1], residual=dxb)
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
    dg, g2 = linear_backward(lin, g_out, dout, need_dw=base_grads)
    grads = {"net.2." + k_: v for k_, v in g2.items()}
    dg = dg[:, :g_out.shape[1]] if dg.shape[1] != g_out.shape[1] else dg
    w, b = geglu._cache_get((dtype,), lambda: packing.pack_geglu_weight(
        geglu.proj.weight.detach().float(), geglu.proj.bias.detach().float(), dtype, ops.k_tile(dtype)))
    pre = ops.linear(x, w, bias=b)                                        # [M, 8C] interleaved (value, gate)
    dpre = ops.geglu_backward(pre, dg.contiguous())
    if base_grads:
        dbi = ops.zeros_f32((dpre.shape[1],), dpre.device)
        dwi = ops.conv2d_wgrad(x, dpre, 1, M, 1, dbias=dbi)[:, :geglu.proj.in_features]
        half = dwi.shape[0] // 2
        # rows are interleaved (value_j, gate_j): back to diffusers' (value | gate) chunk order
        grads["net.0.proj.weight"] = torch.cat([dwi[0::2], dwi[1::2]], 0).contiguous()
        grads["net.0.proj.bias"] = torch.cat([dbi[0::2], dbi[1::2]], 0).contiguous()
        assert grads["net.0.proj.weight"].shape[0] == 2 * half
    wt = geglu._cache_get((dtype, "dgrad"), lambda: ops.pack_dgrad_weights(w, 1))
    dx = ops.conv2d_dgrad(dpre, wt, 1, M, 1, C=w.shape[1])
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


# ----------------------------------------------------------------------------- whole UNet
def _ensure_stats(tok):
    if tok.stats is None:
        tok.stats = ops.new_chsums(tok.B, tok.C, tok.t.device)
        ops.groupnorm_stats(tok.t, tok.B, tok.HW, tok.stats)
    return tok.stats


def transformer2d_backward(tr, x, dout, ctx, Lk, kvs=None, base_grads=True):
    """Backward of ``sd_unet.Transformer2DModel.forward(x, ctx, Lk)`` = proj_out(blocks(proj_in(GroupNorm(x)))) + x:
    returns (dx [M, C], [dctx or dkv per block], grads).  ``kvs``: {id(attn2): precomputed K/V view} or None."""
    B, L = x.B, x.HW
    _ensure_stats(x)
    n = tr.norm(x)
    hin = tr.proj_in(n, stats=False)
    t_in, t = [], hin.t
    for blk in tr.transformer_blocks:
        t_in.append(t)
        kv = None if kvs is None else kvs.get(id(blk.attn2))
        t = blk(t, B, L, ctx, Lk) if kv is None else _block_forward_kv(blk, t, B, L, Lk, kv)
    grads = {}
    dt, g = conv2d_backward(tr.proj_out, x.like(t), dout, need_dw=base_grads)
    grads.update({"proj_out." + k_: v for k_, v in g.items()})
    C = t.shape[1]
    dkvs = []
    for i in reversed(range(len(tr.transformer_blocks))):
        blk = tr.transformer_blocks[i]
        kv = None if kvs is None else kvs.get(id(blk.attn2))
        dt, dkv, g = transformer_block_backward(blk, t_in[i], _cols(dt, C), B, L, ctx, Lk, kv=kv, base_grads=base_grads)
        grads.update({f"transformer_blocks.{i}." + k_: v for k_, v in g.items()})
        dkvs.insert(0, dkv)
    dn, g = conv2d_backward(tr.proj_in, n, _cols(dt, C), need_dw=base_grads)
    grads.update({"proj_in." + k_: v for k_, v in g.items()})
    (dx,), dg, db = ops.groupnorm_backward([x.t], dn, B, L, tr.norm.num_groups, tr.norm.weight.detach().float(),
                                           tr.norm.bias.detach().float(), tr.norm.eps, [x.stats], act="none", dres=dout)
    grads["norm.weight"], grads["norm.bias"] = dg, db
    return dx, dkvs, grads


def _block_forward_kv(blk, h, B, L, Lk, kv):
    from .sd_unet import CtxKV
    return blk(h, B, L, CtxKV(None, {id(blk.attn2): kv}), Lk)


class _Tape:
    """Gradients keyed by the activation tensors of a recorded forward (the tape keeps them alive, so data_ptr() is a
    unique key); a tensor feeding several consumers accumulates through madm_add."""

    def __init__(self):
        self.records = []
        self.g = {}

    def add(self, t, g):
        cur = self.g.get(t.data_ptr())
        self.g[t.data_ptr()] = g if cur is None else ops.add(cur, g.contiguous())

    def pop(self, t):
        return self.g.pop(t.data_ptr(), None)


def unet_forward_recorded(unet, sample, timesteps, ctx, Lk, unet_block_indices, cond_emb=None):
    """``sd_unet.UNet2DConditionModel.forward(sample, timesteps, ctx, Lk, cond_emb, unet_block_indices, 'after')`` run block
    by block while recording every block's INPUT on a tape (block interiors are not kept).  Returns
    (sample_out Tok, tapped features list[Tok], state for :func:`unet_backward_from_state`)."""
    from .sd_unet import CtxKV
    dtype = sample.t.dtype
    B = sample.B
    names = {id(m): n for n, m in unet.named_modules()}
    tape = _Tape()
    rec = tape.records

    # ---- recorded forward (mirrors UNet2DConditionModel.forward) ----
    freqs = ops.timestep_freqs(unet.block_out_channels[0], sample.t.device)
    t_emb = ops.timestep_embedding(timesteps, freqs, dtype)
    te = unet.time_embedding
    e1 = te.linear_1(t_emb)
    a1 = ops.silu(e1)
    res = None if cond_emb is None else ops.cast_from_f32(cond_emb.contiguous(), dtype)
    emb = te.linear_2(a1, residual=res)
    rows = unet._time_rows(emb)
    ctxkv = unet._ctx_kv(ctx)
    kvs = ctxkv.kv if isinstance(ctxkv, CtxKV) else None

    def run_tr(tr, h):
        out = tr(h, ctxkv, Lk)
        rec.append(("tr", tr, h, out))
        return out

    def run_res(r, h, skip=None):
        out = r(h, rows[id(r)], skip=skip)
        rec.append(("res", r, h, skip, out))
        return out

    h = unet.conv_in(sample)
    rec.append(("conv_in", unet.conv_in, sample, h))
    skips = [h]
    for blk in unet.down_blocks:
        for i, r in enumerate(blk.resnets):
            h = run_res(r, h)
            if blk.has_cross_attention:
                h = run_tr(blk.attentions[i], h)
            skips.append(h)
        if blk.downsamplers is not None:
            for d in blk.downsamplers:
                x_in = h
                h = d(h)
                rec.append(("conv", d.conv, x_in, h, False))
            skips.append(h)
    mb = unet.mid_block
    h = run_res(mb.resnets[0], h)
    for attn, r in zip(mb.attentions, mb.resnets[1:]):
        h = run_tr(attn, h)
        h = run_res(r, h)
    idx, tapped = 0, []
    for blk in unet.up_blocks:
        for i, r in enumerate(blk.resnets):
            h = run_res(r, h, skip=skips.pop())
            if blk.has_cross_attention:
                h = run_tr(blk.attentions[i], h)
            if idx in unet_block_indices:
                tapped.append(h)
            idx += 1
        if blk.upsamplers is not None:
            for u in blk.upsamplers:
                x_in = h
                h = u(h)
                rec.append(("conv", u.conv, x_in, h, True))
    out = unet.conv_out(h, norm=unet.conv_norm_out, stats=False)
    st = SimpleNamespace(unet=unet, sample=sample, ctx=ctx, Lk=Lk, cond_emb=cond_emb, dtype=dtype, B=B, names=names,
                         tape=tape, rows=rows, kvs=kvs, emb=emb, e1=e1, a1=a1, t_emb=t_emb, h=h, tapped=tapped)
    return out, tapped, st


def unet_backward_from_state(st, dtaps, dsample=None, base_grads=True, grad_cb=None):
    """Reverse walk over the tape of :func:`unet_forward_recorded` for the gradients ``dtaps`` of the tapped features
    ([M_i, C_i] tensors, same order) and, optionally, ``dsample`` of the final sample ([M, n_pad]); every block's
    backward recomputes its interior.

    Returns {"sample": d latents [M, Cpad], "ctx": d prompt tokens [B*Lk, 768], "cond_emb": f32 [B, 1280] or None,
    "grads": {parameter name: f32 gradient}} -- with ``base_grads=False`` (the reference's LoRA mode) only LoRA A / B,
    the norms' affine parameters and the time-row projections come back.  ``grad_cb(name, tensor)``: called for every
    parameter gradient AS SOON AS its block's backward has run (up blocks first, conv_in last, then the batched K/V and
    time-embedding tails) instead of collecting them -- the hook the overlapped gradient all-reduce hangs on."""
    unet, sample, ctx, Lk, cond_emb, dtype, B, names = st.unet, st.sample, st.ctx, st.Lk, st.cond_emb, st.dtype, st.B, st.names
    tape, rows, kvs, emb, e1, a1, t_emb, h, tapped = st.tape, st.rows, st.kvs, st.emb, st.e1, st.a1, st.t_emb, st.h, st.tapped
    rec = tape.records
    te = unet.time_embedding
    assert len(tapped) == len(dtaps)
    for tk, g in zip(tapped, dtaps):
        tape.add(tk.t, g)

    class _Grads(dict):
        def __setitem__(self, k_, v):
            if grad_cb is not None:
                grad_cb(k_, v)
            else:
                dict.__setitem__(self, k_, v)

    grads = _Grads()
    dctx = None
    drow = {}
    dkv_of = {}

    def put(prefix, g):
        for k_, v in g.items():
            grads[prefix + "." + k_] = v

    # ---- conv_out(silu(conv_norm_out(h))) ----
    if dsample is not None:
        _ensure_stats(h)
        a = unet.conv_norm_out(h, silu=True)
        da, g = conv2d_backward(unet.conv_out, a, dsample, need_dw=base_grads)
        put("conv_out", g)
        no = unet.conv_norm_out
        (dh,), dg, db = ops.groupnorm_backward([h.t], da, B, h.HW, no.num_groups, no.weight.detach().float(),
                                               no.bias.detach().float(), no.eps, [h.stats], act="silu")
        grads["conv_norm_out.weight"], grads["conv_norm_out.bias"] = dg, db
        tape.add(h.t, dh)

    # ---- reverse walk ----
    for r_ in reversed(rec):
        kind = r_[0]
        if kind == "res":
            _, mod, x, skip, out = r_
            dout = tape.pop(out.t)
            if dout is None:
                continue
            dx, dskip, dtr, g = resnet_block_backward(mod, x, dout, temb_row=rows[id(mod)], skip=skip,
                                                      base_grads=base_grads)
            put(names[id(mod)], g)
            drow[id(mod)] = dtr
            tape.add(x.t, dx.t)
            if skip is not None:
                tape.add(skip.t, dskip.t)
        elif kind == "tr":
            _, mod, x, out = r_
            dout = tape.pop(out.t)
            if dout is None:
                continue
            dx, dkvs, g = transformer2d_backward(mod, x, dout, ctx, Lk, kvs=kvs, base_grads=base_grads)
            put(names[id(mod)], g)
            tape.add(x.t, dx)
            for blk, dkv in zip(mod.transformer_blocks, dkvs):
                if kvs is not None:
                    dkv_of[id(blk.attn2)] = dkv
                else:
                    dctx = dkv if dctx is None else ops.add(dctx, dkv)
        elif kind == "conv":
            _, conv, x, out, ups = r_
            dout = tape.pop(out.t)
            if dout is None:
                continue
            dx, g = conv2d_backward(conv, x, dout, upsample=ups, need_dw=base_grads)
            put(names[id(conv)], g)
            tape.add(x.t, dx)
        elif kind == "conv_in":
            _, conv, x, out = r_
            dout = tape.pop(out.t)
            dx, g = conv2d_backward(conv, x, dout, need_dw=base_grads)
            put("conv_in", g)
            tape.add(x.t, dx)
    dsample_in = tape.pop(sample.t)

    # ---- batched K/V projection of the cross-attention layers (no LoRA on to_k / to_v) ----
    if kvs is not None:
        attns = unet._cross_attentions()
        Wall = unet.__dict__["_ctxkv_cache"][dtype][1]
        dall = torch.zeros((ctx.shape[0], Wall.shape[0]), dtype=dtype, device=ctx.device)
        off = 0
        for a_ in attns:
            n = 2 * a_.heads * a_.dim_head
            if id(a_) in dkv_of:
                dall[:, off:off + n] = dkv_of[id(a_)]
            off += n
        dctx = ops.conv2d_dgrad(dall, ops.pack_dgrad_weights(Wall, 1), 1, ctx.shape[0], 1, C=Wall.shape[1])
        if base_grads:
            dW = ops.conv2d_wgrad(ctx, dall, 1, ctx.shape[0], 1)
            off = 0
            for a_ in attns:
                c = a_.heads * a_.dim_head
                grads[names[id(a_)] + ".to_k.weight"] = dW[off:off + c, :a_.to_k.in_features].contiguous()
                grads[names[id(a_)] + ".to_v.weight"] = dW[off + c:off + 2 * c, :a_.to_v.in_features].contiguous()
                off += 2 * c

    # ---- time rows -> time embedding MLP ----
    resnets = unet._resnets()
    _, Wp, _ = unet.__dict__["_temb_cache"][dtype]
    drows = torch.zeros((B, Wp.shape[0]), dtype=torch.float32, device=sample.t.device)
    off = 0
    for r in resnets:
        n = r.time_emb_proj.out_features
        if drow.get(id(r)) is not None:
            drows[:, off:off + n] = drow[id(r)]
        off += n
    drows_c = ops.cast_from_f32(drows, dtype)
    s_emb = ops.silu(emb)
    dWs = ops.conv2d_wgrad(s_emb, drows_c, 1, B, 1)
    dbs = drows.sum(0)
    off = 0
    for r in resnets:
        n = r.time_emb_proj.out_features
        grads[names[id(r)] + ".time_emb_proj.weight"] = dWs[off:off + n, :r.time_emb_proj.in_features].contiguous()
        grads[names[id(r)] + ".time_emb_proj.bias"] = dbs[off:off + n].contiguous()
        off += n
    ds = ops.conv2d_dgrad(drows_c, ops.pack_dgrad_weights(Wp, 1), 1, B, 1, C=Wp.shape[1])
    demb = ops.silu_backward(emb, _cols(ds, emb.shape[1]))
    da1, g = linear_backward(te.linear_2, a1, demb, need_dw=base_grads)
    put("time_embedding.linear_2", g)
    if base_grads:
        de1 = ops.silu_backward(e1, _cols(da1, e1.shape[1]))
        _, g = linear_backward(te.linear_1, t_emb, de1, need_dx=False)
        put("time_embedding.linear_1", g)
    return {"sample": dsample_in, "ctx": dctx, "cond_emb": ops.rows_to_f32(demb) if cond_emb is not None else None,
            "grads": grads}


def unet_backward(unet, sample, timesteps, ctx, Lk, dtaps, unet_block_indices, cond_emb=None, dsample=None,
                  base_grads=True):
    """:func:`unet_forward_recorded` + :func:`unet_backward_from_state` in one call (see there)."""
    _, _, st = unet_forward_recorded(unet, sample, timesteps, ctx, Lk, unet_block_indices, cond_emb=cond_emb)
    return unet_backward_from_state(st, dtaps, dsample=dsample, base_grads=base_grads)
