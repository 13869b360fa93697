"""ORACLE (test infrastructure only): the reference's OWN arithmetic, emulated on the CPU.

The reference never runs the path in fp32: training wraps the step in ``torch.cuda.amp.autocast()``
(/root/reference/engine/train_loop.py:263,277), evaluation does the same (evaluation/evaluator.py:62-66,85), the VAE is
loaded with ``torch_dtype=torch.float16`` and a frozen UNet too (modeling/meta_arch/ldm_diffusers.py:248,253-255).  This
module restates what CUDA autocast does to the oracle modules' ops (ATen autocast_mode.cpp, fp16 policy lists) as a
``TorchFunctionMode`` over CPU tensors that really carry the dtypes the GPU tensors would:

* "lower precision" ops -- conv2d, linear, matmul / bmm, scaled_dot_product_attention: every floating tensor argument is
  rounded to fp16, the product is accumulated in fp32 (what the matrix units do) and the result is rounded to fp16 once;
  SDPA rounds the probabilities to fp16 before the PV product like a flash kernel;
* "fp32" ops -- group_norm, layer_norm, softmax: arguments cast to fp32, fp32 result;
* everything else (adds of the residual stream, SiLU, GELU, concat, nearest upsample) runs in the dtype type promotion
  gives it, on fp16 tensors where the GPU would hold fp16 tensors.

``half_parameters_`` rounds a module's parameters to fp16 values (kept in fp32 storage: the casts above reproduce the
storage type where it matters), as ``from_pretrained(torch_dtype=float16)`` does.

It answers one question (VERDICT r3 "What's weak" 2 / item 5): how far is the reference's own fp16-autocast arithmetic
from the fp32 oracle on the metric's path -- the yardstick for the HIP f16 mode's 1.7e-3 .. 3.5e-3.  tools/precision_autocast.py
runs it; tests/test_oracle.py pins it on the 64 x 64 case.
"""
import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

_LOW = torch.float16


def _h(t):
    return t.to(_LOW) if torch.is_tensor(t) and t.is_floating_point() else t


def _f(t):
    return t.float() if torch.is_tensor(t) and t.is_floating_point() else t


def _sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
    assert attn_mask is None and dropout_p == 0.0 and not is_causal
    q, k, v = _h(q).float(), _h(k).float(), _h(v).float()
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    p = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
    return (p.to(_LOW).float() @ v).to(_LOW)


# ---- block-scaled fp8 (OCP MX, e4m3) for the PV product of attention: what would it cost in accuracy?   -----
def mx_quantize_e4m3(x, dim, block=32):
    """OCP microscaling: along ``dim``, every ``block`` consecutive elements share one power-of-two scale
    2^(floor(log2(amax)) - 8) (E8M0; 8 = the largest e4m3 exponent) and are stored as e4m3 (3 mantissa bits, max 448, saturating)
    -- what v_mfma_scale_f32_16x16x128_f8f6f4 consumes: 32 K-elements and one scale byte per lane.  Returns the dequantised
    values in f32 (the instruction multiplies elements and scales exactly and accumulates in f32)."""
    x = x.float().movedim(dim, -1)
    shp = x.shape
    L = shp[-1]
    pad = (-L) % block
    xb = F.pad(x, (0, pad)).reshape(*shp[:-1], (L + pad) // block, block)
    amax = xb.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(torch.clamp(amax, min=2.0 ** -120))) - 8.0
    scale = torch.exp2(e)
    q = torch.clamp(xb / scale, -448.0, 448.0).to(torch.float8_e4m3fn).float() * scale
    q = torch.where(amax > 0, q, torch.zeros_like(q))
    return q.reshape(*shp[:-1], L + pad)[..., :L].movedim(-1, dim)


def sdpa_fp8_pv(q, k, v, scale=None):
    """scaled_dot_product_attention as the f16 flash kernel computes it, except that the PV product takes block-scaled e4m3
    operands: P (probabilities relative to the running row maximum, <= 1) quantised per (query, 32-key block), V per (32-key
    block, channel); S = QK^T stays f16 x f16 -> f32 (d = 40 does not fill a K = 128 instruction).  The row sum is taken from
    the QUANTISED probabilities (numerator and denominator see the same p, as the ones-column normaliser of attention.hip does)."""
    q, k, v = _h(q).float(), _h(k).float(), _h(v).float()
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    s_ = (q @ k.transpose(-1, -2)) * scale
    p = torch.exp(s_ - s_.amax(dim=-1, keepdim=True))
    pq = mx_quantize_e4m3(p, dim=-1)
    vq = mx_quantize_e4m3(v, dim=-2)
    return ((pq @ vq) / pq.sum(dim=-1, keepdim=True)).to(_LOW)


class CudaAutocastF16(TorchFunctionMode):
    """``with CudaAutocastF16(): module(x)`` -- see the module docstring."""

    LOWER = {F.conv2d, torch.conv2d, F.linear, torch.matmul, torch.bmm, torch.Tensor.matmul, torch.Tensor.__matmul__,
             torch.Tensor.bmm, torch.mm}
    FP32 = {F.group_norm, torch.group_norm, F.layer_norm, torch.layer_norm, F.softmax, torch.softmax, torch.Tensor.softmax}

    fp8_pv_min_keys = None      # set to a key count: self-attention launches with at least that many keys take sdpa_fp8_pv
    fp8_hits = 0

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is F.scaled_dot_product_attention:
            q, k = args[0], args[1]
            if self.fp8_pv_min_keys is not None and k.shape[-2] >= self.fp8_pv_min_keys and q.shape[-2] == k.shape[-2]:
                self.fp8_hits += 1
                return sdpa_fp8_pv(args[0], args[1], args[2], kwargs.get("scale"))
            return _sdpa(*args, **kwargs)
        if func in self.LOWER:
            a = [_h(x).float() if torch.is_tensor(x) and x.is_floating_point() else x for x in args]
            kw = {k: (_h(x).float() if torch.is_tensor(x) and x.is_floating_point() else x) for k, x in kwargs.items()}
            return func(*a, **kw).to(_LOW)
        if func in self.FP32:
            return func(*[_f(x) for x in args], **{k: _f(x) for k, x in kwargs.items()})
        return func(*args, **kwargs)


# ---- Winograd F(2 x 2, 3 x 3) in the 16-bit mode: what would it cost in accuracy?   ----------
_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
_BT = torch.tensor([[1.0, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
_AT = torch.tensor([[1.0, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv3x3_f16(x, w, bias=None, operand_dtype=_LOW):
    """3 x 3 / stride 1 / pad 1 convolution as a 16-bit MFMA kernel would run Winograd F(2 x 2, 3 x 3) (Lavin & Gray 2016):
    the transformed weight U = G g G^T is formed once from the f16 weight (f64) and ROUNDED to the operand type (cached at
    pack time); the transformed input V = B^T d B is formed from the f16 activations in f32 (exact: four signed terms) and
    ROUNDED to the operand type; the 16 per-position products accumulate in f32 (exact products of 16-bit operands); the
    output transform A^T M A runs in f32; one rounding of the result to f16.  The direct form rounds NEITHER operand again:
    products of the stored f16 values accumulate exactly -- that difference is what this function measures."""
    B, C, H, W = x.shape
    N = w.shape[0]
    assert H % 2 == 0 and W % 2 == 0 and tuple(w.shape[1:]) == (C, 3, 3)
    xh = x.to(_LOW).float()
    U = (_G @ w.to(_LOW).double() @ _G.T).float().to(operand_dtype).float()                     # [N, C, 4, 4]
    d = F.pad(xh, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)                                 # [B, C, H/2, W/2, 4, 4]
    V = (_BT @ d @ _BT.T).to(operand_dtype).float()
    th, tw = H // 2, W // 2
    M = torch.einsum("bcijxy,ncxy->bnijxy", V, U)                                               # f32 accumulation over c
    Y = _AT @ M @ _AT.T                                                                         # [B, N, th, tw, 2, 2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, N, H, W)
    if bias is not None:
        y = y + bias.float().view(1, -1, 1, 1)
    return y.to(_LOW)


class CudaAutocastF16Winograd(CudaAutocastF16):
    """CudaAutocastF16 with every eligible convolution (3 x 3, stride 1, pad 1, Cin in ``cins``, maps of at least ``min_hw``
    pixels a side) computed by ``winograd_conv3x3_f16``; ``self.hits`` lists what was replaced.  The SD VAE's 3 x 3 convs have
    128 / 256 / 512 input channels, the UNet's none of these."""

    def __init__(self, cins=(128, 256), min_hw=256, operand_dtype=_LOW):
        super().__init__()
        self.cins, self.min_hw, self.operand_dtype = set(cins), min_hw, operand_dtype
        self.hits = []

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (F.conv2d, torch.conv2d):
            names = ("input", "weight", "bias", "stride", "padding", "dilation", "groups")
            a = dict(zip(names, args))
            a.update(kwargs)
            x, w = a["input"], a["weight"]

            def one(v, d):
                v = a.get(v, d)
                return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
            if (tuple(w.shape[2:]) == (3, 3) and one("stride", 1) == (1, 1) and one("padding", 0) == (1, 1)
                    and one("dilation", 1) == (1, 1) and a.get("groups", 1) == 1 and w.shape[1] in self.cins
                    and min(x.shape[2:]) >= self.min_hw):
                self.hits.append((tuple(x.shape), w.shape[0]))
                return winograd_conv3x3_f16(x, w, a.get("bias"), self.operand_dtype)
        return super().__torch_function__(func, types, args, kwargs)


@torch.no_grad()
def half_parameters_(module):
    """Parameters and buffers rounded to fp16 VALUES (``torch_dtype=torch.float16`` of ldm_diffusers.py:248,253-255)."""
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_floating_point():
            t.copy_(t.to(_LOW).float())
    return module
