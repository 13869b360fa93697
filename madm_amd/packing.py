"""Weight packing: torch/diffusers parameter layouts -> the [N][K] operands of madm_conv2d_fwd.

K is ordered (kh, kw, c) with c running over source 1 then source 2 (channels-last gather order of
the implicit GEMM); every source's channel count is zero-padded to the K-tile (64 bf16 / 32 f32).

The torch statements below define the layouts (and serve CPU tensors: tests, tools); weights that live on the GPU are packed
by one launch of madm_pack_weight / madm_fold_layernorm_pack (ops.pack_weight, ops.fold_layernorm_pack) -- after every
optimizer step every packed operand is re-derived, and as torch ops that was 2 200 launches per training step.
"""
import torch


import os

_TORCH_PACK = bool(int(os.environ.get("MADM_TORCH_PACK", "0")))     # A/B runs: the torch statements on the GPU, as before


def _hip(w):
    return w.is_cuda and w.dtype == torch.float32 and not _TORCH_PACK


def round_up(x, m):
    return (x + m - 1) // m * m


def pack_conv_weight(w, dtype, ktile, splits=None):
    """w: [N, Cin, KH, KW] f32 (nn.Conv2d) -> [N, KH*KW*sum(pad(splits))] of ``dtype``.

    ``splits`` = channel counts of the concatenated sources (default: one source)."""
    N, Cin, KH, KW = w.shape
    if splits is None:
        splits = [Cin]
    assert sum(splits) == Cin
    if _hip(w) and KH * KW <= 9 and len(splits) <= 4:
        from . import ops
        return ops.pack_weight(w, dtype, ktile, splits)
    parts = []
    c0 = 0
    for c in splits:
        part = w[:, c0:c0 + c]
        cp = round_up(c, ktile)
        if cp != c:
            part = torch.nn.functional.pad(part, (0, 0, 0, 0, 0, cp - c))
        parts.append(part)
        c0 += c
    wp = torch.cat(parts, dim=1)                       # [N, Cpad, KH, KW]
    wp = wp.permute(0, 2, 3, 1).contiguous()           # [N, KH, KW, Cpad]
    return wp.reshape(N, -1).to(dtype).contiguous()


def unpack_conv_weight_grad(dw, Cin, KH, KW, ktile, splits=None):
    """Inverse of :func:`pack_conv_weight` for the f32 gradient ops.conv2d_wgrad returns:
    [N, KH*KW*sum(pad(splits))] -> [N, Cin, KH, KW] (the gradient of the padding channels is dropped).  A VIEW of ``dw``
    for a single source (the consumer adds it into the parameter's gradient: one strided add instead of copy + add)."""
    N = dw.shape[0]
    if splits is None:
        splits = [Cin]
    cpads = [round_up(c, ktile) for c in splits]
    g = dw.reshape(N, KH, KW, sum(cpads))
    if len(splits) == 1:
        return g[..., :splits[0]].permute(0, 3, 1, 2)
    parts, p0 = [], 0
    for c, cp in zip(splits, cpads):
        parts.append(g[..., p0:p0 + c])
        p0 += cp
    return torch.cat(parts, dim=-1).permute(0, 3, 1, 2)


def pack_linear_weight(w, dtype, ktile):
    """w: [N, K] (nn.Linear) -> [N, pad(K)]."""
    N, K = w.shape
    if _hip(w):
        from . import ops
        return ops.pack_weight(w, dtype, ktile)
    Kp = round_up(K, ktile)
    if Kp != K:
        w = torch.nn.functional.pad(w, (0, Kp - K))
    return w.to(dtype).contiguous()


def pack_geglu_weight(w, b, dtype, ktile):
    """diffusers GEGLU.proj: Linear(C, 8C) whose output is chunked (value | gate).  Rows are
    interleaved (value_j, gate_j) so one lane's 4 consecutive outputs hold two complete pairs."""
    N, K = w.shape
    half = N // 2
    bi = torch.stack([b[:half], b[half:]], dim=1).reshape(N)
    if _hip(w):
        from . import ops
        return ops.pack_weight(w, dtype, ktile, interleave=True), bi.float().contiguous()
    wi = torch.stack([w[:half], w[half:]], dim=1).reshape(N, K)
    return pack_linear_weight(wi, dtype, ktile), bi.float().contiguous()


def fold_layernorm(w, b, gamma, beta, dtype, ktile, interleave=False):
    """LayerNorm folded into the Linear that consumes it (madm_conv2d_args.ln_colsum):  Linear(LN(x)) =
    rstd (x W'^T - mean colsum(W')) + (W beta + b)  with  W' = W * gamma  (column k scaled by gamma[k]).
    w [N, K] f32 (rows already in the kernel's order, e.g. GEGLU-interleaved), b [N] f32 or None ->
    (packed W' [N, pad(K)] of ``dtype``, bias' f32 [N], colsum f32 [N] of the ROUNDED W' so that the mean correction
    cancels exactly what the MFMAs accumulate).  ``interleave``: w / b are in diffusers GEGLU order (values | gates) and
    the result is in the kernel's (value_j, gate_j) row order."""
    N, K = w.shape
    assert K % ktile == 0, "the folded LayerNorm normalises over K = C: no channel padding"
    if _hip(w):
        from . import ops
        return ops.fold_layernorm_pack(w, b, gamma, beta, dtype, interleave)
    if interleave:
        half = N // 2
        w = torch.stack([w[:half], w[half:]], dim=1).reshape(N, K)
        b = None if b is None else torch.stack([b[:half], b[half:]], dim=1).reshape(N)
    wp = (w.double() * gamma.double()[None, :])
    bias = (w.double() @ beta.double())
    if b is not None:
        bias = bias + b.double()
    wq = wp.float().to(dtype)
    return wq.contiguous(), bias.float().contiguous(), wq.double().sum(dim=1).float().contiguous()
