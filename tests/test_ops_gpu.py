"""Per-kernel parity on the GPU: every C-ABI entry point against the matching torch-CPU fp32 op
(SURVEY.md 8c pin K7).  f32 mode must agree to fp32 rounding; bf16 mode is compared with the same
op evaluated in fp32 on bf16-rounded operands (only accumulation order and output rounding differ)."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import to_tokens, from_tokens, rel_err, bf16_round

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2}


def _q(x, dtype):
    return bf16_round(x) if dtype == torch.bfloat16 else (x.to(torch.float16).float() if dtype == torch.float16 else x)


def _gen(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g)


CONV_CASES = [
    # name, B, Cin(list), H, W, Cout, KH, stride, pad_mode, upsample, tile, splitk
    ("3x3_s1", 2, [64], 8, 8, 64, 3, 1, "same", False, 0, 1),
    ("3x3_s1_bigtile", 2, [128], 16, 16, 256, 3, 1, "same", False, 1, 1),
    ("3x3_s1_tile2", 1, [64], 12, 20, 320, 3, 1, "same", False, 2, 1),
    ("3x3_s1_ragged", 2, [64], 5, 7, 320, 3, 1, "same", False, 0, 1),
    ("3x3_s2_unet", 2, [64], 16, 16, 128, 3, 2, "same", False, 0, 1),
    ("3x3_s2_vae_asym", 2, [64], 16, 16, 128, 3, 2, "asym", False, 0, 1),
    ("3x3_upsample", 2, [64], 6, 6, 64, 3, 1, "same", True, 0, 1),
    ("3x3_concat", 2, [128, 64], 8, 8, 192, 3, 1, "same", False, 0, 1),
    ("3x3_splitk", 2, [256], 4, 4, 128, 3, 1, "same", False, 3, 4),
    ("3x3_autosplit", 2, [640], 2, 2, 64, 3, 1, "same", False, 0, None),
    ("1x1", 2, [128], 8, 8, 64, 1, 1, "none", False, 0, 1),
    ("3x3_tiny_cin", 1, [3], 16, 16, 128, 3, 1, "same", False, 0, 1),
    ("3x3_tiny_cout", 2, [64], 8, 8, 4, 3, 1, "same", False, 0, 1),
    # LDS halo-tile kernel (tile codes 4 = x128 channels, 5 = x64), full / ragged patches, concat, split-K
    ("halo128", 2, [128], 16, 32, 256, 3, 1, "same", False, 4, 1),
    ("halo64", 2, [64], 16, 16, 320, 3, 1, "same", False, 5, 1),
    ("halo_ragged", 1, [64], 13, 21, 192, 3, 1, "same", False, 4, 1),
    ("halo_concat", 2, [128, 64], 8, 16, 128, 3, 1, "same", False, 5, 1),
    ("halo_splitk", 2, [256], 8, 16, 64, 3, 1, "same", False, 5, 2),
    ("halo_auto", 1, [64], 64, 64, 128, 3, 1, "same", False, 0, None),
    # 6 = igemm 64x64 with the 8-deep prefetch ring (K long enough to wrap it, and a split K shorter than it)
    ("deep_igemm64", 2, [640], 8, 8, 320, 1, 1, "same", False, 6, 1),
    ("deep_igemm64_splitk", 2, [1280], 4, 4, 128, 3, 1, "same", False, 6, 3),
    # 7 / 8 = LDS-DMA igemm (buffer_load ... lds, 4- / 3-slot ring): padding rows must land as zeros, ragged M / N,
    # K shorter and longer than the ring, concat, stride 2, upsample, split-K
    ("glds64_3x3", 2, [64], 8, 8, 64, 3, 1, "same", False, 7, 1),
    ("glds64_ragged", 2, [64], 5, 7, 320, 3, 1, "same", False, 7, 1),
    ("glds64_1x1_longk", 2, [1280], 8, 8, 320, 1, 1, "none", False, 7, 1),
    ("glds64_1x1_onetile", 2, [64], 8, 8, 64, 1, 1, "none", False, 7, 1),
    ("glds64_concat_s2", 2, [128, 64], 16, 16, 192, 3, 2, "same", False, 7, 1),
    ("glds64_upsample", 2, [64], 6, 6, 64, 3, 1, "same", True, 7, 1),
    ("glds64_splitk", 2, [256], 4, 4, 128, 3, 1, "same", False, 7, 4),
    ("glds128_3x3", 1, [64], 12, 20, 320, 3, 1, "same", False, 8, 1),
    ("glds128_asym_s2", 2, [64], 16, 16, 128, 3, 2, "asym", False, 8, 1),
    ("glds128_1x1_longk_splitk", 2, [640], 8, 8, 640, 1, 1, "none", False, 8, 2),
    ("glds128_tiny_cout", 2, [64], 8, 8, 4, 3, 1, "same", False, 8, 1),
    # 14 / 15 = LDS-DMA igemm 128 x 128 (3-slot ring, one block per CU / 2-slot ring, two blocks per CU): ragged M and N,
    # K shorter and longer than the ring, gathers with padding, concat + stride 2, upsample, split-K
    ("glds128sq_1x1_longk", 2, [1280], 8, 16, 320, 1, 1, "none", False, 14, 1),
    ("glds128sq_3x3_ragged", 1, [64], 13, 21, 192, 3, 1, "same", False, 14, 1),
    ("glds128sq_concat_s2_splitk", 2, [128, 64], 16, 16, 192, 3, 2, "same", False, 14, 2),
    ("glds128sq_onetile", 2, [64], 8, 8, 64, 1, 1, "none", False, 14, 1),
    ("glds128sq2_1x1_longk", 2, [1280], 8, 16, 320, 1, 1, "none", False, 15, 1),
    ("glds128sq2_3x3_ragged", 1, [64], 13, 21, 192, 3, 1, "same", False, 15, 1),
    ("glds128sq2_upsample_splitk", 2, [128], 6, 6, 64, 3, 1, "same", True, 15, 3),
    ("glds128sq2_onetile", 2, [64], 8, 8, 64, 1, 1, "none", False, 15, 1),
    ("glds64d_1x1_longk", 2, [1280], 8, 16, 320, 1, 1, "none", False, 16, 1),
    ("glds64d_3x3_ragged", 1, [64], 13, 21, 192, 3, 1, "same", False, 16, 1),
    ("glds64d_concat_s2_splitk", 2, [128, 64], 16, 16, 192, 3, 2, "same", False, 16, 2),
    ("glds128x64d_1x1_longk", 2, [1280], 8, 16, 320, 1, 1, "none", False, 17, 1),
    ("glds128x64d_3x3_ragged", 1, [64], 13, 21, 192, 3, 1, "same", False, 17, 1),
    ("glds128x64d_upsample_splitk", 2, [128], 6, 6, 64, 3, 1, "same", True, 17, 3),
    ("glds64s_1x1", 2, [320], 8, 8, 320, 1, 1, "none", False, 11, 1),
    ("glds64s_ragged_3x3_splitk", 2, [128], 5, 7, 192, 3, 1, "same", False, 11, 2),
    # 9 / 10 = halo conv with LDS-DMA weights (three-slot ring, single halo buffer)
    ("halodma128", 2, [128], 16, 32, 256, 3, 1, "same", False, 9, 1),
    ("halodma64", 2, [64], 16, 16, 320, 3, 1, "same", False, 10, 1),
    ("halodma_ragged", 1, [64], 13, 21, 192, 3, 1, "same", False, 9, 1),
    ("halodma_concat", 2, [128, 64], 8, 16, 128, 3, 1, "same", False, 10, 1),
    ("halodma_splitk", 2, [256], 8, 16, 64, 3, 1, "same", False, 10, 2),
    ("halodma_longk", 2, [640], 8, 16, 128, 3, 1, "same", False, 9, 1),
    # 12 = 16 x 16-patch halo conv with the nearest-2x upsample folded into its halo gather (Upsample2D), full / ragged
    ("h16_upsample", 2, [64], 16, 16, 128, 3, 1, "same", True, 12, 1),
    ("h16_upsample_ragged_concat", 1, [128, 64], 9, 13, 192, 3, 1, "same", True, 12, 1),
    # split-K slices of the 16 x 16-patch kernel (its own lean epilogue instantiation): K = 9 x 192 in 3 slices, ragged patches, N = 192
    ("h16_splitk3_ragged", 2, [128, 64], 19, 35, 192, 3, 1, "same", False, 12, 3),
    ("h16_splitk2_upsample", 1, [128], 16, 16, 128, 3, 1, "same", True, 12, 2),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d(cuda, dtype, case):
    from madm_amd import ops, packing
    from madm_amd._lib import lib
    name, B, cins, H, W, Cout, KH, stride, pad_mode, ups, tile, splitk = case
    kt = ops.k_tile(dtype)
    xs = [_q(_gen((B, c, H, W), 10 + i), dtype) for i, c in enumerate(cins)]
    Cin = sum(cins)
    w = _q(_gen((Cout, Cin, KH, KH), 3) / math.sqrt(Cin * KH * KH), dtype)
    bias = _gen((Cout,), 4)
    rowvec = _gen((B, Cout), 5)
    x = torch.cat(xs, 1)
    if ups:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    if pad_mode == "same":
        ref = F.conv2d(x, w, None, stride=stride, padding=KH // 2)
        pad_t = pad_l = KH // 2
    elif pad_mode == "asym":
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, None, stride=stride, padding=0)
        pad_t = pad_l = 0
    else:
        ref = F.conv2d(x, w, None, stride=stride, padding=0)
        pad_t = pad_l = 0
    OH, OW = ref.shape[2], ref.shape[3]
    res = _q(_gen((B, Cout, OH, OW), 6), dtype)
    ref = ref + bias[None, :, None, None] + rowvec[:, :, None, None] + res

    cpads = [packing.round_up(c, kt) for c in cins]
    toks = [to_tokens(xi, dtype, cp) for xi, cp in zip(xs, cpads)]
    wp = packing.pack_conv_weight(w, dtype, kt, splits=cins).cuda()
    lib.madm_debug_set_conv_tile(tile)
    try:
        out = ops.conv2d(toks[0], wp, B, H, W, N=Cout, x2=toks[1] if len(toks) > 1 else None, KH=KH, KW=KH,
                         stride=stride, pad_t=pad_t, pad_l=pad_l, OH=OH, OW=OW, upsample=ups,
                         bias=bias.cuda(), rowvec=rowvec.cuda(), residual=to_tokens(res, dtype),
                         splitk=splitk)
        torch.cuda.synchronize()
    finally:
        lib.madm_debug_set_conv_tile(0)
    got = from_tokens(out, B, OH, OW)
    e, l2 = rel_err(got, ref)
    assert e < TOL[dtype], f"{name}: max rel err {e:.3e} l2 {l2:.3e}"



H16_CASES = [
    # name, B, Cin(list), H, W, Cout, fuse_gn (None / act), relu epilogue, residual, stats
    ("full", 2, [128], 32, 32, 128, None, False, True, True),
    ("ragged_n192", 1, [64], 21, 37, 192, None, False, True, True),
    ("concat", 2, [128, 64], 16, 32, 128, None, False, False, True),
    ("relu_nores", 2, [64], 16, 16, 256, None, True, False, False),
    ("gn_silu", 2, [128], 32, 16, 128, "silu", False, True, True),
    ("gn_relu_ragged", 1, [64], 19, 23, 64, "relu", False, False, True),
    ("gn_none_concat_straddle", 2, [1280, 640], 16, 16, 64, None, False, False, False),
    ("gn_silu_concat_straddle", 2, [1280, 640], 16, 16, 64, "silu", False, True, True),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", H16_CASES, ids=[c[0] for c in H16_CASES])
def test_conv3x3_h16(cuda, dtype, case):
    """The 16 x 16-patch halo conv (tile code 12: the dominant kernel of the metric) forced on small maps: full and ragged
    patches, N not a multiple of the block width, two sources, every epilogue option (bias, time row, residual, ReLU,
    fused output statistics) on both store paths (16-bit: LDS-transposed 16-byte stores; f32: direct), and the fused
    GroupNorm(+SiLU / ReLU) input pass incl. groups straddling the source boundary and the packed-f16 form."""
    from madm_amd import ops, packing
    from madm_amd._lib import lib
    name, B, cins, H, W, Cout, fuse, relu, with_res, with_stats = case
    kt = ops.k_tile(dtype)
    Cin = sum(cins)
    xs = [_q(_gen((B, c, H, W), 30 + i) * (1.0 + 0.5 * i) + 0.25 * i, dtype) for i, c in enumerate(cins)]
    w = _q(_gen((Cout, Cin, 3, 3), 3) / math.sqrt(Cin * 9), dtype)
    bias, rowvec = _gen((Cout,), 4), _gen((B, Cout), 5)
    h = torch.cat(xs, 1)
    gn = None
    toks = [to_tokens(x, dtype) for x in xs]
    if fuse is not None or name.startswith("gn_"):
        gamma, beta = 1.0 + 0.2 * _gen((Cin,), 6), 0.3 * _gen((Cin,), 7)
        h = F.group_norm(h, 32, gamma, beta, eps=1e-5)
        h = F.silu(h) if fuse == "silu" else (F.relu(h) if fuse == "relu" else h)
        sts = []
        for t in toks:
            st = torch.zeros((B, t.shape[1], 2), dtype=torch.float64, device="cuda")
            ops.groupnorm_stats(t, B, H * W, st)
            sts.append(st)
        gn = (sts, gamma.cuda(), beta.cuda(), 32, 1e-5, fuse if fuse is not None else False)
    ref = F.conv2d(h, w, None, padding=1) + bias[None, :, None, None] + rowvec[:, :, None, None]
    res = _q(_gen((B, Cout, H, W), 8), dtype) if with_res else None
    if res is not None:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    wp = packing.pack_conv_weight(w, dtype, kt, splits=cins).cuda()
    st_out = torch.zeros((B, Cout, 2), device="cuda", dtype=torch.float64) if with_stats else None
    lib.madm_debug_set_conv_tile(12)
    try:
        out = ops.conv2d(toks[0], wp, B, H, W, N=Cout, x2=toks[1] if len(toks) > 1 else None, KH=3, KW=3, pad_t=1, pad_l=1,
                         bias=bias.cuda(), rowvec=rowvec.cuda(), residual=None if res is None else to_tokens(res, dtype),
                         epilogue=ops.EPI_RELU if relu else ops.EPI_NONE, stats=st_out, gn=gn, splitk=1)
        torch.cuda.synchronize()
    finally:
        lib.madm_debug_set_conv_tile(0)
    tol = {torch.float32: 3e-5, torch.bfloat16: 2e-2, torch.float16: 3e-3}[dtype]
    e, l2 = rel_err(from_tokens(out, B, H, W), ref)
    assert e < tol, f"{name}: max rel err {e:.3e} l2 {l2:.3e}"
    if st_out is not None:
        sums = st_out.float().cpu()
        stol = {torch.float32: 1e-4, torch.bfloat16: 2e-2, torch.float16: 3e-3}[dtype]
        assert rel_err(sums[..., 0], ref.sum((2, 3)))[0] < stol
        assert rel_err(sums[..., 1], (ref ** 2).sum((2, 3)))[0] < stol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 16, 128), (1, 37, 133), (3, 5, 300)], ids=["full_seg", "ragged", "multi_seg"])
def test_stem_conv3x3(cuda, dtype, shape):
    """madm_stem_conv3x3: normalise + the VAE stem conv (3 -> 128, 3x3 / pad 1, zero padding AFTER the normalisation) +
    bias + fused output statistics + range probe, straight from the f32 NCHW image."""
    from madm_amd import ops
    B, H, W = shape
    img = torch.rand((B, 3, H, W), generator=torch.Generator().manual_seed(5))
    w = _gen((128, 3, 3, 3), 6) / math.sqrt(27)
    bias = _gen((128,), 7)
    mean, std = 0.5, 0.5
    ref = F.conv2d((img - mean) / std, w, bias, padding=1)
    wT = w.permute(2, 3, 1, 0).reshape(27, 128).contiguous().cuda()
    st = torch.zeros((B, 128, 2), dtype=torch.float64, device="cuda")
    mm = torch.tensor([float("inf"), float("-inf")], device="cuda")
    out = ops.stem_conv3x3(img.cuda(), wT, bias.cuda(), dtype, mean, std, stats=st, minmax=mm)
    torch.cuda.synchronize()
    if dtype != torch.float32:
        # 16-bit modes run the stem on the matrix pipe: the normalised image and the weights are rounded to the dtype and
        # accumulated in f32 -- what conv_in does under the reference's autocast -- so THAT is the exact statement;
        # against the unrounded conv only the operand rounding shows
        e0 = rel_err(from_tokens(out, B, H, W), ref)[0]
        assert e0 < {torch.bfloat16: 2e-2, torch.float16: 3e-3}[dtype], f"{e0:.3e}"
        ref = F.conv2d(_q((img - mean) / std, dtype), _q(w, dtype), bias, padding=1)
    e, l2 = rel_err(from_tokens(out, B, H, W), ref)
    assert e < {torch.float32: 2e-6, torch.bfloat16: 6e-3, torch.float16: 8e-4}[dtype], f"{e:.3e} {l2:.3e}"
    sums = st.float().cpu()
    assert rel_err(sums[..., 0], ref.sum((2, 3)))[0] < 1e-4 and rel_err(sums[..., 1], (ref ** 2).sum((2, 3)))[0] < 1e-4
    if dtype != torch.float32:   # the exact-f32 FMA kernel stays selectable (A/B runs) and keeps its tighter statement
        import os
        os.environ["MADM_STEM_KERNEL"] = "1"
        try:
            out1 = ops.stem_conv3x3(img.cuda(), wT, bias.cuda(), dtype, mean, std)
        finally:
            del os.environ["MADM_STEM_KERNEL"]
        ref0 = F.conv2d((img - mean) / std, w, bias, padding=1)
        assert rel_err(from_tokens(out1, B, H, W), ref0)[0] < {torch.bfloat16: 6e-3, torch.float16: 8e-4}[dtype]
    lo, hi = mm.tolist()
    x = (img - mean) / std
    assert abs(lo - float(x.min())) < 1e-6 and abs(hi - float(x.max())) < 1e-6

@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_linear_geglu(cuda, dtype):
    from madm_amd import ops, packing
    kt = ops.k_tile(dtype)
    M, C = 70, 64
    x = _q(_gen((M, C), 1), dtype)
    w = _q(_gen((8 * C, C), 2) / math.sqrt(C), dtype)
    b = _gen((8 * C,), 3)
    h = F.linear(x, w, b)
    val, gate = h.chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    wp, bp = packing.pack_geglu_weight(w, b, dtype, kt)
    out = ops.linear(x.to(dtype).cuda(), wp.cuda(), bias=bp.cuda(), epilogue=ops.EPI_GEGLU)
    e, l2 = rel_err(out.float().cpu(), ref)
    assert e < TOL[dtype], f"geglu: {e:.3e} {l2:.3e}"


LN_FOLD_CASES = [
    # name, M, C, N, tile (0 = tuned / heuristic), geglu, residual
    ("qkv_64sq", 300, 320, 960, 0, False, False),
    ("q_glds64", 512, 1280, 1280, 7, False, True),
    ("q_glds64s_ragged", 130, 640, 640, 11, False, True),
    ("geglu_glds128", 256, 320, 2560, 8, True, False),
    ("geglu_reg128x64", 200, 640, 1024, 2, True, False),
    ("reg128x128", 256, 320, 256, 1, False, False),
    ("glds128sq_qkv", 300, 320, 960, 14, False, False),
    ("glds128sq_geglu", 256, 640, 1024, 14, True, False),
    ("glds128sq2_q_res", 200, 1280, 640, 15, False, True),
    ("glds128sq2_fullchip_geglu", 8192, 320, 2560, 15, True, False),
    ("glds64d_qkv", 300, 320, 960, 16, False, False),
    ("glds64d_geglu", 256, 640, 1024, 16, True, False),
    ("glds128x64d_q_res", 200, 1280, 640, 17, False, True),
    ("glds128x64d_geglu", 1024, 320, 2560, 17, True, False),
    ("reg64x64", 96, 64, 128, 3, False, True),
    ("reg64x64d", 70, 128, 64, 6, False, False),
    # tile 13 = A-stationary kernel (igemm_apanel.hip): resident row panel (BM = 128 / 64 / 32 by K), LayerNorm in place
    ("apanel128_qkv", 300, 320, 960, 13, False, False),
    ("apanel128_geglu_manytiles", 1000, 320, 2560, 13, True, False),
    ("apanel64_q", 200, 640, 640, 13, False, False),
    ("apanel64_geglu", 130, 640, 1024, 13, True, False),
    ("apanel32_qkv", 96, 1280, 3840, 13, False, False),
    ("apanel32_ragged_n", 70, 1280, 72, 13, False, False),
    ("apanel128_onetile", 128, 64, 64, 13, False, False),
    # full-chip grids (several workgroups per CU, MFMA loops beside epilogues / LayerNorm passes of other workgroups): the
    # packed-FP32 observation of csrc/Makefile showed only there -- a dropped mean is an error of O(1), far above the gate
    ("apanel_fullchip_geglu", 8192, 320, 2560, 13, True, False),
    ("glds128_fullchip_qkv", 8192, 320, 960, 8, False, False),
    ("reg128x64_fullchip_geglu", 2048, 640, 5120, 2, True, False),
    ("glds64_fullchip_q", 8192, 320, 320, 11, False, True),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", LN_FOLD_CASES, ids=[c[0] for c in LN_FOLD_CASES])
def test_linear_with_folded_layernorm(cuda, dtype, case):
    """Linear(LayerNorm(x)) as ONE GEMM (madm_conv2d_args.ln_colsum, packing.fold_layernorm): the three norms of diffusers'
    BasicTransformerBlock in front of to_q/k/v, attn2.to_q and the GEGLU projection.  Rows with a large mean / spread test
    the cancellation in  rstd (x W'^T - mean colsum W');  reference = torch LayerNorm + linear in f32 on the rounded x."""
    from madm_amd import ops, packing
    from madm_amd._lib import lib
    name, M, C, N, tile, geglu, with_res = case
    kt = ops.k_tile(dtype)
    x = _gen((M, C), 1) * (0.5 + 2.0 * torch.rand((M, 1), generator=torch.Generator().manual_seed(7))) \
        + 3.0 * _gen((M, 1), 8)                                            # per-row scale and offset
    x = _q(x, dtype)
    gamma, beta = 1.0 + 0.3 * _gen((C,), 2), 0.2 * _gen((C,), 3)
    w = _gen((N, C), 4) / math.sqrt(C)
    b = _gen((N,), 5)
    ref = F.linear(F.layer_norm(x, (C,), gamma, beta, 1e-5), w, b)
    if geglu:
        val, gate = ref.chunk(2, dim=-1)
        ref = val * F.gelu(gate)
        half = N // 2
        wk = torch.stack([w[:half], w[half:]], dim=1).reshape(N, C)
        bk = torch.stack([b[:half], b[half:]], dim=1).reshape(N)
    else:
        wk, bk = w, b
    res = _q(_gen((M, N), 6), dtype) if with_res else None
    if with_res:
        ref = ref + res
    if tile == 13 and dtype == torch.float32 and C > 640:
        pytest.skip("f32 rows of more than 2560 bytes do not fit the 32-row panel (the 64 x 64 kernels serve them)")
    wp, bp, cs = packing.fold_layernorm(wk, bk, gamma, beta, dtype, kt)
    lib.madm_debug_set_conv_tile(tile)
    ops.PROFILE = []
    try:
        out = ops.linear(x.to(dtype).cuda(), wp.cuda(), bias=bp.cuda(), ln=(cs.cuda(), 1e-5),
                         epilogue=ops.EPI_GEGLU if geglu else ops.EPI_NONE,
                         residual=None if res is None else res.to(dtype).cuda())
        torch.cuda.synchronize()
        kernel = ops.PROFILE[0][0]
    finally:
        ops.PROFILE = None
        lib.madm_debug_set_conv_tile(0)
    assert (tile != 13) or kernel.startswith("igemm_apanel"), kernel
    if tile == 13:      # and the same kernel without the LayerNorm: plain Linear (+ GEGLU)
        lib.madm_debug_set_conv_tile(13)
        try:
            wq = packing.pack_linear_weight(wk, dtype, kt)
            o2 = ops.linear(x.to(dtype).cuda(), wq.cuda(), bias=bk.cuda(), epilogue=ops.EPI_GEGLU if geglu else ops.EPI_NONE)
            torch.cuda.synchronize()
        finally:
            lib.madm_debug_set_conv_tile(0)
        r2 = F.linear(x, _q(wk, dtype), bk)
        if geglu:
            r2 = r2[:, 0::2] * F.gelu(r2[:, 1::2])
        e2, _ = rel_err(o2.float().cpu(), r2)
        assert e2 < TOL.get(dtype, 2.5e-3), f"{name} (no LayerNorm): {e2:.3e}"
    e, l2 = rel_err(out.float().cpu(), ref)
    # f16 / bf16: W' = W * gamma is rounded once more than the unfused path's weights; same order as the GEMM's own error
    tol = {torch.float32: 3e-5, torch.bfloat16: 1.5e-2, torch.float16: 2.5e-3}[dtype]
    assert e < tol, f"{name}: max rel err {e:.3e} l2 {l2:.3e}"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 8, 8), (2, 320, 16, 16), (1, 128, 33, 17), (2, 1280, 2, 2)])
@pytest.mark.parametrize("silu", [False, True])
def test_groupnorm(cuda, dtype, shape, silu):
    from madm_amd import ops
    B, C, H, W = shape
    x = _q(_gen(shape, 1) * 2.0 + 0.5, dtype)
    gamma = _gen((C,), 2)
    beta = _gen((C,), 3)
    ref = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    if silu:
        ref = F.silu(ref)
    out = ops.groupnorm(to_tokens(x, dtype), B, H * W, 32, gamma.cuda(), beta.cuda(), 1e-5, silu=silu)
    e, l2 = rel_err(from_tokens(out, B, H, W), ref)
    assert e < (1e-5 if dtype == torch.float32 else 1e-2), f"{e:.3e} {l2:.3e}"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 1280, 640, 4, 4), (2, 320, 320, 8, 8), (1, 64, 128, 5, 3)])
def test_groupnorm_two_sources(cuda, dtype, shape):
    """GroupNorm over the skip concatenation [x | skip] without materialising the concat input; groups
    may straddle the source boundary (1280 + 640 channels: 60 per group)."""
    from madm_amd import ops
    B, C1, C2, H, W = shape
    x1 = _q(_gen((B, C1, H, W), 1) * 1.5 + 0.3, dtype)
    x2 = _q(_gen((B, C2, H, W), 2) * 0.7 - 0.2, dtype)
    gamma = _gen((C1 + C2,), 3)
    beta = _gen((C1 + C2,), 4)
    ref = F.silu(F.group_norm(torch.cat([x1, x2], 1), 32, gamma, beta, eps=1e-5))
    out = ops.groupnorm([to_tokens(x1, dtype), to_tokens(x2, dtype)], B, H * W, 32, gamma.cuda(), beta.cuda(), 1e-5,
                        silu=True)
    e, l2 = rel_err(from_tokens(out, B, H, W), ref)
    assert e < (1e-5 if dtype == torch.float32 else 1e-2), f"{e:.3e} {l2:.3e}"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 64, 16, 16, 128, 0, 1), (2, 64, 8, 8, 320, 3, 1), (2, 256, 4, 4, 64, 3, 4),
                                  (3, 64, 3, 3, 64, 0, 1), (2, 640, 2, 2, 128, 0, None)],
                         ids=["tile_auto", "tile64", "splitk", "straddle", "autosplit"])
def test_conv_fused_groupnorm_stats(cuda, dtype, case):
    """The conv epilogue's per-(image, channel) sum / sum of squares equal those of its own output, and a
    GroupNorm fed with them equals F.group_norm of the conv result."""
    from madm_amd import ops, packing
    from madm_amd._lib import lib
    B, Cin, H, W, Cout, tile, splitk = case
    kt = ops.k_tile(dtype)
    x = _q(_gen((B, Cin, H, W), 1), dtype)
    w = _q(_gen((Cout, Cin, 3, 3), 2) / math.sqrt(Cin * 9), dtype)
    bias = _gen((Cout,), 3)
    wp = packing.pack_conv_weight(w, dtype, kt).cuda()
    st = torch.zeros((B, Cout, 2), device="cuda", dtype=torch.float64)
    lib.madm_debug_set_conv_tile(tile)
    try:
        out = ops.conv2d(to_tokens(x, dtype), wp, B, H, W, N=Cout, KH=3, KW=3, pad_t=1, pad_l=1, bias=bias.cuda(),
                         stats=st, splitk=splitk)
    finally:
        lib.madm_debug_set_conv_tile(0)
    o = out.float().cpu().reshape(B, H * W, Cout)
    conv_ref = F.conv2d(x, w, bias, padding=1)
    sums = st.float().cpu()
    assert rel_err(sums[..., 0], conv_ref.sum((2, 3)))[0] < (1e-4 if dtype == torch.float32 else 2e-2)
    assert rel_err(sums[..., 1], (conv_ref ** 2).sum((2, 3)))[0] < (1e-4 if dtype == torch.float32 else 2e-2)
    gamma, beta = _gen((Cout,), 4), _gen((Cout,), 5)
    y = ops.groupnorm(out, B, H * W, 32, gamma.cuda(), beta.cuda(), 1e-5, silu=False, stats=[st])
    ref = F.group_norm(o.permute(0, 2, 1).reshape(B, Cout, H, W), 32, gamma, beta, eps=1e-5)
    e, l2 = rel_err(from_tokens(y, B, H, W), ref)
    assert e < (2e-5 if dtype == torch.float32 else 1e-2), f"{e:.3e} {l2:.3e}"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, [128], 16, 32, 128, True, 4), (2, [320], 16, 16, 320, True, 5),
                                  (1, [64], 11, 19, 64, False, 5), (2, [128, 64], 8, 16, 192, True, 5),
                                  (2, [1280, 640], 8, 16, 64, True, 5), (2, [128], 16, 32, 128, True, 9),
                                  (2, [320], 16, 16, 320, True, 10), (2, [1280, 640], 8, 16, 64, True, 10),
                                  (1, [64], 11, 19, 64, False, 9)],
                         ids=["x128", "x64", "ragged_affine", "concat", "concat_straddle", "dma128", "dma64",
                              "dma_concat_straddle", "dma_ragged_affine"])
def test_conv_fused_groupnorm_input(cuda, dtype, case):
    """conv3x3(silu(GroupNorm([x | skip]))) with the normalisation folded into the conv's LDS halo load
    (channel sums -> group mean / rstd in the conv's own prologue) equals the unfused torch composition,
    including the zero padding applied AFTER the activation."""
    from madm_amd import ops, packing
    from madm_amd._lib import lib
    B, cins, H, W, Cout, act, tile = case
    kt = ops.k_tile(dtype)
    Cin = sum(cins)
    xs = [_q(_gen((B, c, H, W), 20 + i) * (1.0 + i) + 0.5 * i, dtype) for i, c in enumerate(cins)]
    gamma, beta = 1.0 + 0.2 * _gen((Cin,), 2), 0.3 * _gen((Cin,), 3)
    w = _q(_gen((Cout, Cin, 3, 3), 4) / math.sqrt(Cin * 9), dtype)
    bias = _gen((Cout,), 5)
    h = F.group_norm(torch.cat(xs, 1), 32, gamma, beta, eps=1e-5)
    if act:
        h = F.silu(h)
    ref = F.conv2d(h, w, bias, padding=1)
    toks = [to_tokens(x, dtype) for x in xs]
    sts = []
    for t in toks:
        st = torch.zeros((B, t.shape[1], 2), dtype=torch.float64, device="cuda")
        ops.groupnorm_stats(t, B, H * W, st)
        sts.append(st)
    wp = packing.pack_conv_weight(w, dtype, kt, splits=cins).cuda()
    lib.madm_debug_set_conv_tile(tile)
    try:
        out = ops.conv2d(toks[0], wp, B, H, W, N=Cout, x2=toks[1] if len(toks) > 1 else None, KH=3, KW=3, pad_t=1,
                         pad_l=1, bias=bias.cuda(), gn=(sts, gamma.cuda(), beta.cuda(), 32, 1e-5, act))
    finally:
        lib.madm_debug_set_conv_tile(0)
    e, l2 = rel_err(from_tokens(out, B, H, W), ref)
    # bf16: the fused path rounds the activated input once to bf16 exactly like the unfused one
    assert e < (3e-5 if dtype == torch.float32 else 2e-2), f"{e:.3e} {l2:.3e}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_weight_packing_kernels_equal_the_torch_statements(cuda, dtype):
    """madm_pack_weight / madm_fold_layernorm_pack (what the modules use for weights that live on the GPU) produce the
    layouts packing.py states in torch (run here on the CPU copies): bit for bit, except the f64-accumulated folded bias."""
    from madm_amd import ops, packing
    kt = ops.k_tile(dtype)
    g = torch.Generator().manual_seed(11)
    # conv weights: one source, padded source, two / three concatenated sources (UNet up blocks), 1x1, N not a multiple of 32
    for (N, splits, k) in [(64, [64], 3), (36, [4], 3), (320, [640, 320], 3), (68, [100, 28, 64], 1), (128, [300], 1)]:
        w = torch.randn((N, sum(splits), k, k), generator=g)
        ref = packing.pack_conv_weight(w, dtype, kt, splits)
        got = packing.pack_conv_weight(w.cuda(), dtype, kt, splits)
        assert got.is_cuda and got.shape == ref.shape and torch.equal(got.cpu(), ref), (N, splits, k)
    for (N, K) in [(320, 320), (77, 768), (10, 100)]:                      # linear, K padded to the tile
        w = torch.randn((N, K), generator=g)
        assert torch.equal(packing.pack_linear_weight(w.cuda(), dtype, kt).cpu(), packing.pack_linear_weight(w, dtype, kt))
    w, b = torch.randn((2560, 320), generator=g), torch.randn((2560,), generator=g)      # GEGLU row interleave
    rw, rb = packing.pack_geglu_weight(w, b, dtype, kt)
    gw, gb = packing.pack_geglu_weight(w.cuda(), b.cuda(), dtype, kt)
    assert torch.equal(gw.cpu(), rw) and torch.equal(gb.cpu(), rb)
    # folded LayerNorm: plain rows (fused Q/K/V), no bias (cross-attention to_q), GEGLU order
    for (N, K, with_b, inter) in [(960, 320, True, False), (640, 640, False, False), (2560, 320, True, True)]:
        w = torch.randn((N, K), generator=g) / K ** 0.5
        b = torch.randn((N,), generator=g) if with_b else None
        gamma, beta = 1 + 0.2 * torch.randn((K,), generator=g), 0.3 * torch.randn((K,), generator=g)
        rW, rB, rC = packing.fold_layernorm(w, b, gamma, beta, dtype, kt, interleave=inter)
        gW, gB, gC = packing.fold_layernorm(w.cuda(), None if b is None else b.cuda(), gamma.cuda(), beta.cuda(), dtype, kt,
                                            interleave=inter)
        assert torch.equal(gW.cpu(), rW) and torch.equal(gC.cpu(), rC), (N, K)
        assert rel_err(gB.cpu(), rB)[0] < 1e-6, (N, K)


POST_GN_CASES = [
    # name, B, Cin, H, W, Cout, splitk, tile (0 = table / heuristic), time row, expected "applied"
    ("unet16_1280", 2, 64, 16, 16, 1280, 4, 0, True, True),       # 40 KB of LDS per group
    ("unet32_640", 2, 192, 32, 32, 640, 3, 0, True, True),        # 80 KB; halo kernel, three channel chunks in 16-bit modes
    # 92 KB AFTER the 80 KB launch in the same process: the dynamic-LDS attribute is raised once per device and must
    # cover the kernel's maximum, not the first launch's size (a 24 x 24 map of a 768-px eval image)
    ("unet24_1280_after_80kb", 1, 64, 24, 24, 1280, 4, 0, False, True),
    ("unet8_1280_halo", 1, 128, 8, 8, 1280, 2, 9, False, True),   # the halo kernel's split-K (whole channel chunks)
    ("ragged_map", 3, 64, 5, 7, 128, 5, 3, True, True),
    ("ten_channel_groups", 2, 64, 8, 8, 320, 9, 3, True, True),   # 8-byte units (the 16-byte path needs N / groups % 4 == 0)
    ("clamped_to_one_slab", 2, 64, 8, 8, 128, 4, 9, False, None),  # halo kernel: Cin = 64 is one chunk in 16-bit modes
    ("no_splitk", 2, 64, 16, 16, 128, 1, 0, True, False),
    ("group_too_large", 1, 64, 64, 64, 320, 2, 0, False, False),  # 160 KB
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", POST_GN_CASES, ids=[c[0] for c in POST_GN_CASES])
def test_conv_post_groupnorm_on_the_splitk_reduction(cuda, dtype, case):
    """conv2d(post_gn=...) == silu(GroupNorm(conv(x) + bias + time row)) (diffusers ResnetBlock2D conv1 -> norm2 ->
    nonlinearity): applied by the split-K reduction when the launch has one and the group fits LDS; ``applied`` False
    otherwise, with the raw conv output and its statistics as without post_gn."""
    from madm_amd import ops, packing
    from madm_amd._lib import lib
    name, B, Cin, H, W, Cout, splitk, tile, with_row, expect = case
    kt = ops.k_tile(dtype)
    x = _q(_gen((B, Cin, H, W), 1), dtype)
    w = _q(_gen((Cout, Cin, 3, 3), 2) / math.sqrt(Cin * 9), dtype)
    bias = _gen((Cout,), 3)
    row = (0.5 * _gen((B, Cout), 6)) if with_row else None
    gamma, beta = 1.0 + 0.2 * _gen((Cout,), 4), 0.3 * _gen((Cout,), 5)
    wp = packing.pack_conv_weight(w, dtype, kt).cuda()
    st = torch.zeros((B, Cout, 2), device="cuda", dtype=torch.float64)
    lib.madm_debug_set_conv_tile(tile)
    try:
        out, applied = ops.conv2d(to_tokens(x, dtype), wp, B, H, W, N=Cout, KH=3, KW=3, pad_t=1, pad_l=1, bias=bias.cuda(),
                                  rowvec=None if row is None else row.cuda(), stats=st, splitk=splitk,
                                  post_gn=(gamma.cuda(), beta.cuda(), 32, 1e-5, True))
        raw = ops.conv2d(to_tokens(x, dtype), wp, B, H, W, N=Cout, KH=3, KW=3, pad_t=1, pad_l=1, bias=bias.cuda(),
                         rowvec=None if row is None else row.cuda(), splitk=splitk)
    finally:
        lib.madm_debug_set_conv_tile(0)
    if expect is None:
        expect = dtype == torch.float32     # K tile of 32 elements: two chunks of Cin = 64 only in f32
    assert applied == expect, (name, applied)
    conv_ref = F.conv2d(x, w, bias, padding=1)
    if row is not None:
        conv_ref = conv_ref + row[:, :, None, None]
    if not applied:
        assert torch.equal(out, raw)
        assert rel_err(st.float().cpu()[..., 0], conv_ref.sum((2, 3)))[0] < (1e-4 if dtype == torch.float32 else 2e-2)
        return
    assert float(st.abs().max()) == 0.0     # the statistics buffer is not touched on the merged path
    ref = F.silu(F.group_norm(conv_ref, 32, gamma, beta, eps=1e-5))
    e, l2 = rel_err(from_tokens(out, B, H, W), ref)
    assert e < (3e-5 if dtype == torch.float32 else 2e-2), f"{e:.3e} {l2:.3e}"
    # and against the two-launch composition on the same slabs: the merged kernel normalises the f32 sums, the
    # composition the values rounded to the compute dtype -- equal up to that rounding (f32: up to summation order)
    st2 = torch.zeros((B, Cout, 2), device="cuda", dtype=torch.float64)
    ops.groupnorm_stats(raw, B, H * W, st2)
    two = ops.groupnorm(raw, B, H * W, 32, gamma.cuda(), beta.cuda(), 1e-5, silu=True, stats=[st2])
    e2, _ = rel_err(out.float().cpu(), two.float().cpu())
    assert e2 < (2e-5 if dtype == torch.float32 else 2e-2), f"{e2:.3e}"


WGRAD_CASES = [
    # name, B, Cin(list), H, W, Cout, KH, stride, pad_mode, upsample, splitm
    ("3x3_s1", 2, [64], 8, 8, 64, 3, 1, "same", False, 0),
    ("3x3_s1_wide", 2, [128], 16, 32, 256, 3, 1, "same", False, 0),
    ("3x3_s1_ragged", 1, [64], 5, 7, 320, 3, 1, "same", False, 1),
    ("3x3_s1_sliced", 2, [64], 12, 20, 68, 3, 1, "same", False, 7),
    ("3x3_s2_unet", 2, [64], 16, 16, 128, 3, 2, "same", False, 0),
    ("3x3_s2_vae_asym", 2, [64], 16, 16, 128, 3, 2, "asym", False, 3),
    ("3x3_upsample", 2, [64], 6, 6, 64, 3, 1, "same", True, 0),
    ("3x3_concat", 2, [128, 64], 8, 8, 192, 3, 1, "same", False, 0),
    ("3x3_tiny_cin", 1, [4], 16, 16, 128, 3, 1, "same", False, 0),
    ("3x3_tiny_cout", 2, [64], 8, 8, 4, 3, 1, "same", False, 0),
    ("1x1", 2, [128], 8, 8, 64, 1, 1, "none", False, 0),
    ("linear_lora", 1, [320, 64], 77, 1, 320, 1, 1, "none", False, 0),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv2d_wgrad(cuda, dtype, case):
    """madm_conv2d_wgrad equals torch autograd's weight gradient of the same conv (the reference's backward is
    loss.backward(), engine/train_loop.py:203-217), for every geometry the forward kernel supports; a second call
    accumulates."""
    from madm_amd import ops, packing
    name, B, cins, H, W, Cout, KH, stride, pad_mode, ups, splitm = case
    kt = ops.k_tile(dtype)
    xs = [_q(_gen((B, c, H, W), 30 + i), dtype) for i, c in enumerate(cins)]
    Cin = sum(cins)
    w = (_gen((Cout, Cin, KH, KH), 3) / math.sqrt(Cin * KH * KH)).requires_grad_(True)
    x = torch.cat(xs, 1)
    if ups:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    if pad_mode == "same":
        y = F.conv2d(x, w, None, stride=stride, padding=KH // 2)
        pad_t = pad_l = KH // 2
    elif pad_mode == "asym":
        y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, None, stride=stride, padding=0)
        pad_t = pad_l = 0
    else:
        y = F.conv2d(x, w, None, stride=stride, padding=0)
        pad_t = pad_l = 0
    OH, OW = y.shape[2], y.shape[3]
    dy = _q(_gen(tuple(y.shape), 7), dtype)
    y.backward(dy)
    ref = w.grad

    cpads = [packing.round_up(c, kt) for c in cins]
    toks = [to_tokens(xi, dtype, cp) for xi, cp in zip(xs, cpads)]
    db = torch.zeros(dy.shape[1], device="cuda")
    dw = ops.conv2d_wgrad(toks[0], to_tokens(dy, dtype), B, H, W, x2=toks[1] if len(toks) > 1 else None, KH=KH, KW=KH,
                          stride=stride, pad_t=pad_t, pad_l=pad_l, OH=OH, OW=OW, upsample=ups, splitm=splitm, dbias=db)
    torch.cuda.synchronize()
    # the bias gradient gathered by the same launch: column sums of dout over all pixels
    eb, _ = rel_err(db.cpu(), dy.sum((0, 2, 3)))
    assert eb < 2e-5, f"{name}: bias gradient {eb:.3e}"
    got = packing.unpack_conv_weight_grad(dw.cpu(), Cin, KH, KH, kt, splits=cins)
    e, l2 = rel_err(got, ref)
    # same operands, f32 accumulation in both modes: only the summation order differs
    assert e < 2e-5, f"{name}: max rel err {e:.3e} l2 {l2:.3e}"
    # the gradient of the padding channels is exactly zero
    assert float(dw.abs().sum()) == pytest.approx(float(got.abs().sum()), rel=1e-5)
    ops.conv2d_wgrad(toks[0], to_tokens(dy, dtype), B, H, W, x2=toks[1] if len(toks) > 1 else None, KH=KH, KW=KH,
                     stride=stride, pad_t=pad_t, pad_l=pad_l, OH=OH, OW=OW, upsample=ups, splitm=splitm, dw=dw)
    e2, _ = rel_err(packing.unpack_conv_weight_grad(dw.cpu(), Cin, KH, KH, kt, splits=cins), 2 * ref)
    assert e2 < 2e-5, f"{name}: accumulation {e2:.3e}"


DGRAD_CASES = [
    # name, B, Cin(list), H, W, Cout, KH
    ("3x3", 2, [64], 8, 8, 128, 3),
    ("3x3_ragged", 1, [128], 5, 7, 64, 3),
    ("3x3_concat", 2, [128, 64], 8, 16, 192, 3),
    ("1x1", 2, [128], 8, 8, 64, 1),
    ("linear", 1, [320], 77, 1, 1280, 1),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", DGRAD_CASES, ids=[c[0] for c in DGRAD_CASES])
def test_conv2d_dgrad(cuda, dtype, case):
    """The forward kernel on dout with madm_pack_dgrad_weights' transposed / tap-reversed weights equals torch
    autograd's input gradient of a stride-1 conv / linear (+ the gradient arriving through a skip connection)."""
    from madm_amd import ops, packing
    name, B, cins, H, W, Cout, KH = case
    kt = ops.k_tile(dtype)
    Cin = sum(cins)
    x = _gen((B, Cin, H, W), 40).requires_grad_(True)
    w = _q(_gen((Cout, Cin, KH, KH), 3) / math.sqrt(Cin * KH * KH), dtype)
    y = F.conv2d(x, w, None, padding=KH // 2)
    dy = _q(_gen(tuple(y.shape), 8), dtype)
    skip = _q(_gen((B, Cin, H, W), 9), dtype)
    y.backward(dy)
    ref = x.grad + skip

    cpads = [packing.round_up(c, kt) for c in cins]
    wp = packing.pack_conv_weight(w, dtype, kt, splits=cins).cuda()
    wt = ops.pack_dgrad_weights(wp, KH * KH)
    assert tuple(wt.shape) == (sum(cpads), KH * KH * Cout)
    skip_tok = torch.cat([to_tokens(s_, dtype, cp) for s_, cp in zip(torch.split(skip, cins, 1), cpads)], 1)
    din = ops.conv2d_dgrad(to_tokens(dy, dtype), wt, B, H, W, C=sum(cpads), KH=KH, KW=KH, pad_t=KH // 2,
                           pad_l=KH // 2, residual=skip_tok)
    torch.cuda.synchronize()
    got = from_tokens(din, B, H, W)
    parts, p0 = [], 0
    for c, cp in zip(cins, cpads):
        parts.append(got[:, p0:p0 + c])
        p0 += cp
    e, l2 = rel_err(torch.cat(parts, 1), ref)
    assert e < TOL[dtype], f"{name}: max rel err {e:.3e} l2 {l2:.3e}"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, [128], 16, 16, "silu"), (2, [320], 8, 8, "none"), (1, [64], 11, 19, "relu"),
                                  (2, [128, 64], 8, 16, "silu"), (2, [1280, 640], 4, 6, "silu")],
                         ids=["silu", "plain_cpg10", "relu_ragged", "concat", "concat_straddle"])
def test_groupnorm_backward(cuda, dtype, case):
    """madm_groupnorm_bwd_sums / _apply equal torch autograd through act(F.group_norm(cat(xs))) for dx of every source,
    dgamma and dbeta (groups may straddle the boundary of a two-source concat)."""
    from madm_amd import ops
    B, cins, H, W, act = case
    Cin = sum(cins)
    xs = [_q(_gen((B, c, H, W), 50 + i) * (1.0 + i) + 0.5 * i, dtype).requires_grad_(True) for i, c in enumerate(cins)]
    gamma = (1.0 + 0.2 * _gen((Cin,), 2)).requires_grad_(True)
    beta = (0.3 * _gen((Cin,), 3)).requires_grad_(True)
    y = F.group_norm(torch.cat(xs, 1), 32, gamma, beta, eps=1e-5)
    y = F.silu(y) if act == "silu" else (F.relu(y) if act == "relu" else y)
    dy = _q(_gen(tuple(y.shape), 4), dtype)
    y.backward(dy)
    toks = [to_tokens(x.detach(), dtype) for x in xs]
    sts = []
    for t in toks:
        st = torch.zeros((B, t.shape[1], 2), dtype=torch.float64, device="cuda")
        ops.groupnorm_stats(t, B, H * W, st)
        sts.append(st)
    dxs, dg, db = ops.groupnorm_backward(toks, to_tokens(dy, dtype), B, H * W, 32, gamma.detach().cuda(),
                                         beta.detach().cuda(), 1e-5, sts, act=act)
    torch.cuda.synchronize()
    tol = 3e-5 if dtype == torch.float32 else 1.5e-2
    for x, dx in zip(xs, dxs):
        e, l2 = rel_err(from_tokens(dx, B, H, W), x.grad)
        assert e < tol, f"dx {e:.3e} {l2:.3e}"
    assert rel_err(dg.cpu(), gamma.grad)[0] < tol
    assert rel_err(db.cpu(), beta.grad)[0] < tol


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(77, 320), (1024, 640), (9, 1280), (300, 2560)], ids=["c320", "c640", "c1280", "c2560"])
def test_layernorm_backward(cuda, dtype, case):
    """madm_layernorm_bwd equals torch autograd through F.layer_norm (dx, dgamma, dbeta); a second call accumulates
    the parameter gradients."""
    from madm_amd import ops
    M, C = case
    if dtype == torch.float32 and C > 1280:
        pytest.skip("f32 rows are limited to 1280 channels (as the forward kernel)")
    x = _q(_gen((M, C), 60) * 1.5 + 0.3, dtype).requires_grad_(True)
    gamma = (1.0 + 0.2 * _gen((C,), 2)).requires_grad_(True)
    beta = (0.3 * _gen((C,), 3)).requires_grad_(True)
    y = F.layer_norm(x, (C,), gamma, beta, eps=1e-5)
    dy = _q(_gen((M, C), 5), dtype)
    y.backward(dy)
    xd, dyd = x.detach().to(dtype).cuda(), dy.to(dtype).cuda()
    dx, dg, db = ops.layernorm_backward(xd, dyd, gamma.detach().cuda(), 1e-5)
    torch.cuda.synchronize()
    tol = 3e-5 if dtype == torch.float32 else 1.5e-2
    assert rel_err(dx.float().cpu(), x.grad)[0] < tol
    assert rel_err(dg.cpu(), gamma.grad)[0] < tol
    assert rel_err(db.cpu(), beta.grad)[0] < tol
    ops.layernorm_backward(xd, dyd, gamma.detach().cuda(), 1e-5, dgamma=dg, dbeta=db)
    assert rel_err(dg.cpu(), 2 * gamma.grad)[0] < tol


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 128, 0, 128, 8, 16, True), (2, 64, 0, 128, 8, 8, True), (1, 128, 64, 64, 6, 10, True),
                                  (2, 128, 0, 128, 8, 8, False)],
                         ids=["identity_shortcut", "conv_shortcut", "skip_concat", "no_time_row"])
def test_resnet_block_backward(cuda, dtype, case):
    """backward.resnet_block_backward (GroupNorm / conv weight + data gradients / shortcut / time row composed from the
    C-ABI kernels, activations recomputed) equals torch autograd through diffusers' ResnetBlock2D arithmetic
    (SURVEY.md 8a row a6')."""
    from madm_amd import backward
    from madm_amd.nn import Tok
    from madm_amd.sd_unet import ResnetBlock2D
    B, C1, C2, Cout, H, W, with_temb = case
    Cin = C1 + C2
    blk = ResnetBlock2D(Cin, Cout).cuda()
    ref = {}
    for i, (name, p) in enumerate(blk.named_parameters()):
        if "norm" in name:
            v = (1.0 + 0.2 * _gen(tuple(p.shape), 100 + i)) if name.endswith("weight") else 0.3 * _gen(tuple(p.shape), 100 + i)
        elif name.endswith("weight"):
            v = _q(_gen(tuple(p.shape), 100 + i) / math.sqrt(p[0].numel()), dtype)
        else:
            v = 0.1 * _gen(tuple(p.shape), 100 + i)
        p.data.copy_(v)
        ref[name] = v.clone().requires_grad_(True)
    x = _q(_gen((B, C1, H, W), 1), dtype).requires_grad_(True)
    skip = _q(_gen((B, C2, H, W), 2) * 1.5 + 0.2, dtype).requires_grad_(True) if C2 else None
    temb_row = (0.5 * _gen((B, Cout), 3)).requires_grad_(True) if with_temb else None
    xin = x if skip is None else torch.cat([x, skip], 1)
    a1 = F.silu(F.group_norm(xin, 32, ref["norm1.weight"], ref["norm1.bias"], eps=1e-5))
    h = F.conv2d(a1, ref["conv1.weight"], ref["conv1.bias"], padding=1)
    if with_temb:
        h = h + temb_row[:, :, None, None]
    a2 = F.silu(F.group_norm(h, 32, ref["norm2.weight"], ref["norm2.bias"], eps=1e-5))
    out = F.conv2d(a2, ref["conv2.weight"], ref["conv2.bias"], padding=1)
    out = out + (xin if Cin == Cout else F.conv2d(xin, ref["conv_shortcut.weight"], ref["conv_shortcut.bias"]))
    dout = _q(_gen(tuple(out.shape), 4), dtype)
    out.backward(dout)

    xt = Tok(to_tokens(x.detach(), dtype), B, H, W)
    st = Tok(to_tokens(skip.detach(), dtype), B, H, W) if skip is not None else None
    # the forward of the module under test equals the reference (sanity of the set-up)
    got_out = blk(xt, temb_row=temb_row.detach().cuda() if with_temb else None, skip=st)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert rel_err(from_tokens(got_out.t, B, H, W), out.detach())[0] < tol
    dx, dskip, dtemb, grads = backward.resnet_block_backward(
        blk, xt, to_tokens(dout, dtype), temb_row=temb_row.detach().cuda() if with_temb else None, skip=st)
    torch.cuda.synchronize()
    assert rel_err(from_tokens(dx.t, B, H, W), x.grad)[0] < tol
    if skip is not None:
        assert rel_err(from_tokens(dskip.t, B, H, W), skip.grad)[0] < tol
    if with_temb:
        assert rel_err(dtemb.cpu(), temb_row.grad)[0] < tol
    else:
        assert dtemb is None
    # time_emb_proj belongs to the UNet's stacked time-row GEMM: its gradient flows through dtemb_row
    assert set(grads) == {n for n in ref if not n.startswith("time_emb_proj")}, (sorted(grads), sorted(ref))
    for name, g in grads.items():
        e, l2 = rel_err(g.float().cpu(), ref[name].grad)
        assert e < tol, f"{name}: {e:.3e} {l2:.3e}"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [("s1", 64, 128, 8, 12, 1, 1, False, False), ("s2_unet", 64, 64, 16, 16, 2, 1, False, False),
                                  ("s2_vae_asym", 128, 128, 16, 12, 2, 0, True, False),
                                  ("upsample", 64, 64, 6, 10, 1, 1, False, True), ("1x1", 128, 64, 8, 8, 1, 0, False, False)],
                         ids=lambda c: c[0])
def test_conv_module_backward(cuda, dtype, case):
    """backward.conv2d_backward on the nn.Conv2d twin: weight, bias and data gradients of the stride-1, the two
    stride-2 (Downsample2D) and the nearest-2x upsample (Upsample2D) geometries equal torch autograd."""
    from madm_amd import backward
    from madm_amd.nn import Tok, Conv2d
    name, Cin, Cout, H, W, stride, padding, asym, ups = case
    k = 1 if name == "1x1" else 3
    B = 2
    conv = Conv2d(Cin, Cout, k, stride=stride, padding=padding, asym_pad=asym).cuda()
    w = _q(_gen((Cout, Cin, k, k), 1) / math.sqrt(Cin * k * k), dtype).requires_grad_(True)
    bias = (0.1 * _gen((Cout,), 2)).requires_grad_(True)
    conv.weight.data.copy_(w.detach())
    conv.bias.data.copy_(bias.detach())
    x = _q(_gen((B, Cin, H, W), 3), dtype).requires_grad_(True)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    if asym:
        xin = F.pad(xin, (0, 1, 0, 1))
    y = F.conv2d(xin, w, bias, stride=stride, padding=padding)
    dy = _q(_gen(tuple(y.shape), 4), dtype)
    y.backward(dy)
    xt = Tok(to_tokens(x.detach(), dtype), B, H, W)
    assert tuple(conv.out_hw(H, W, ups)) == tuple(y.shape[2:])
    dx, grads = backward.conv2d_backward(conv, xt, to_tokens(dy, dtype), upsample=ups)
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    assert rel_err(from_tokens(dx, B, H, W), x.grad)[0] < tol
    assert rel_err(grads["weight"].cpu(), w.grad)[0] < 2e-5
    assert rel_err(grads["bias"].cpu(), bias.grad)[0] < 2e-5


def test_silu_backward(cuda):
    from madm_amd import ops
    x = _gen((64, 1280), 1).requires_grad_(True)
    dy = _gen((64, 1280), 2)
    F.silu(x).backward(dy)
    dx = ops.silu_backward(x.detach().cuda(), dy.cuda())
    assert rel_err(dx.cpu(), x.grad)[0] < 1e-5


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 64, 320, 8, 40, 77, True), (1, 100, 640, 8, 80, 77, True),
                                  (2, 64, 320, 8, 40, 77, False)], ids=["c320_lora", "c640_ragged_lora", "c320_plain"])
def test_transformer_block_backward(cuda, dtype, case):
    """backward.transformer_block_backward (LayerNorm / fused QKV + LoRA projections / attention / GEGLU feed-forward
    gradients composed from the C-ABI kernels, everything recomputed from the block input) equals torch autograd
    through diffusers' BasicTransformerBlock arithmetic with peft LoRA on to_q / to_k / to_v / to_out.0
    (the transformer and LoRA rows)."""
    from madm_amd import backward
    from madm_amd.sd_unet import BasicTransformerBlock, LoraLinear
    B, L, C, H, D, Lk, lora = case
    blk = BasicTransformerBlock(C, H, D, 768)
    if lora:
        for attn in (blk.attn1, blk.attn2):
            for nm in ("to_q", "to_k", "to_v"):
                setattr(attn, nm, LoraLinear(getattr(attn, nm)))
            attn.to_out[0] = LoraLinear(attn.to_out[0])
        for m in blk.modules():
            if isinstance(m, LoraLinear):
                m.update_layer("default", 8, 16)
                m._active_adapter = ["default"]
    blk = blk.cuda()
    ref = {}
    for i, (name, p) in enumerate(blk.named_parameters()):
        if "norm" in name:
            v = (1.0 + 0.2 * _gen(tuple(p.shape), 200 + i)) if name.endswith("weight") else 0.3 * _gen(tuple(p.shape), 200 + i)
        elif name.endswith("weight"):
            v = _q(_gen(tuple(p.shape), 200 + i) / math.sqrt(p.shape[1]), dtype)
        else:
            v = 0.1 * _gen(tuple(p.shape), 200 + i)
        p.data.copy_(v)
        ref[name] = v.clone().requires_grad_(True)
    h = _q(_gen((B * L, C), 1), dtype).requires_grad_(True)
    ctx = _q(0.5 * _gen((B * Lk, 768), 2), dtype).requires_grad_(True)

    def lin(x, pre):
        if lora:
            y = F.linear(x, ref[pre + ".base_layer.weight"], ref.get(pre + ".base_layer.bias"))
            return y + F.linear(F.linear(x, ref[pre + ".lora_A.default.weight"]), ref[pre + ".lora_B.default.weight"]) * 2.0
        return F.linear(x, ref[pre + ".weight"], ref.get(pre + ".bias"))

    def attn(pre, x, kvsrc, Lkv):
        q = lin(x, pre + ".to_q").reshape(B, L, H, D).transpose(1, 2)
        k = lin(kvsrc, pre + ".to_k").reshape(B, Lkv, H, D).transpose(1, 2)
        v = lin(kvsrc, pre + ".to_v").reshape(B, Lkv, H, D).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, scale=D ** -0.5).transpose(1, 2).reshape(B * L, H * D)
        return lin(o, pre + ".to_out.0")

    n1 = F.layer_norm(h, (C,), ref["norm1.weight"], ref["norm1.bias"], eps=1e-5)
    h1 = h + attn("attn1", n1, n1, L)
    n2 = F.layer_norm(h1, (C,), ref["norm2.weight"], ref["norm2.bias"], eps=1e-5)
    h2 = h1 + attn("attn2", n2, ctx, Lk)
    n3 = F.layer_norm(h2, (C,), ref["norm3.weight"], ref["norm3.bias"], eps=1e-5)
    val, gate = F.linear(n3, ref["ff.net.0.proj.weight"], ref["ff.net.0.proj.bias"]).chunk(2, dim=-1)
    out = h2 + F.linear(val * F.gelu(gate), ref["ff.net.2.weight"], ref["ff.net.2.bias"])
    dout = _q(_gen((B * L, C), 3), dtype)
    out.backward(dout)

    hd, cd = h.detach().to(dtype).cuda(), ctx.detach().to(dtype).cuda()
    got_out = blk(hd, B, L, cd, Lk)
    tol = 2e-4 if dtype == torch.float32 else 4e-2
    assert rel_err(got_out.float().cpu(), out.detach())[0] < tol
    dh, dctx, grads = backward.transformer_block_backward(blk, hd, dout.to(dtype).cuda(), B, L, cd, Lk)
    torch.cuda.synchronize()
    assert rel_err(dh.float().cpu(), h.grad)[0] < tol
    assert rel_err(dctx.float().cpu()[:, :768], ctx.grad)[0] < tol
    assert set(grads) == set(ref), (sorted(set(grads) ^ set(ref)))
    for name, g in grads.items():
        e, l2 = rel_err(g.float().cpu(), ref[name].grad)
        assert e < tol, f"{name}: {e:.3e} {l2:.3e}"


def test_geglu_backward(cuda):
    from madm_amd import ops
    M, N = 64, 1280
    pre = _gen((M, 2 * N), 1).requires_grad_(True)
    dout = _gen((M, N), 2)
    (pre[:, 0::2] * F.gelu(pre[:, 1::2])).backward(dout)
    dpre = ops.geglu_backward(pre.detach().cuda(), dout.cuda())
    assert rel_err(dpre.cpu(), pre.grad)[0] < 1e-5


def test_groupnorm_finalize_utility(cuda):
    """madm_groupnorm_finalize (stand-alone form of what the fused conv does in its prologue): x * scale + shift
    equals GroupNorm(x) for a two-source concat whose groups straddle the boundary."""
    from madm_amd import ops
    B, cins, H, W = 2, [1280, 640], 4, 6
    xs = [_gen((B, c, H, W), 30 + i) * (1.0 + i) + 0.5 * i for i, c in enumerate(cins)]
    C = sum(cins)
    gamma, beta = 1.0 + 0.2 * _gen((C,), 2), 0.3 * _gen((C,), 3)
    ref = F.group_norm(torch.cat(xs, 1), 32, gamma, beta, eps=1e-5)
    sts = []
    for x in xs:
        t = to_tokens(x, torch.float32)
        st = torch.zeros((B, t.shape[1], 2), dtype=torch.float64, device="cuda")
        ops.groupnorm_stats(t, B, H * W, st)
        sts.append(st)
    scale, shift = ops.groupnorm_finalize(sts, B, H * W, 32, gamma.cuda(), beta.cuda(), 1e-5)
    got = torch.cat(xs, 1) * scale.cpu()[:, :, None, None] + shift.cpu()[:, :, None, None]
    e, l2 = rel_err(got, ref)
    assert e < 1e-5, f"{e:.3e} {l2:.3e}"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("MC", [(70, 320), (16, 640), (5, 1280)])
def test_layernorm(cuda, dtype, MC):
    from madm_amd import ops
    M, C = MC
    x = _q(_gen((M, C), 1) * 3.0 - 1.0, dtype)
    gamma = _gen((C,), 2)
    beta = _gen((C,), 3)
    ref = F.layer_norm(x, (C,), gamma, beta, eps=1e-5)
    out = ops.layernorm(x.to(dtype).cuda(), gamma.cuda(), beta.cuda(), 1e-5)
    e, l2 = rel_err(out.float().cpu(), ref)
    assert e < (1e-5 if dtype == torch.float32 else 1e-2), f"{e:.3e} {l2:.3e}"


ATTN_CASES = [
    # B, H, Lq, Lk, D
    (2, 8, 64, 64, 40),
    (1, 8, 100, 100, 40),
    (2, 8, 16, 16, 80),
    (2, 8, 4, 4, 160),
    (2, 8, 64, 77, 40),     # cross attention over the 77-token prompt
    (2, 8, 16, 77, 160),
    (1, 1, 64, 64, 512),    # VAE mid block
    (1, 2, 200, 333, 64),
    (1, 8, 1, 1, 160),
    (1, 8, 2048, 2048, 40),  # long self-attention maps: two query groups per wave (128-query blocks, Q staged in KV buffer 1)
    (1, 2, 2100, 2100, 40),  # ... with ragged query / key tails
]


@pytest.mark.parametrize("dtype", DTYPES + [torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", ATTN_CASES, ids=[f"B{c[0]}H{c[1]}q{c[2]}k{c[3]}d{c[4]}" for c in ATTN_CASES])
def test_attention(cuda, dtype, case):
    from madm_amd import ops
    B, H, Lq, Lk, D = case
    q = _q(_gen((B, Lq, H, D), 1), dtype)
    k = _q(_gen((B, Lk, H, D), 2), dtype)
    v = _q(_gen((B, Lk, H, D), 3), dtype)
    scale = D ** -0.5
    ref = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale=scale)
    ref = ref.transpose(1, 2).reshape(B * Lq, H * D)
    # q,k,v as column views of one fused buffer (the layout the QKV projection produces)
    if Lq == Lk:
        fused = torch.cat([q.reshape(B * Lq, H * D), k.reshape(B * Lk, H * D), v.reshape(B * Lk, H * D)], 1)
        fused = fused.to(dtype).cuda()
        C = H * D
        qd, kd, vd = fused[:, :C], fused[:, C:2 * C], fused[:, 2 * C:]
    else:
        qd = q.reshape(B * Lq, H * D).to(dtype).cuda()
        kv = torch.cat([k.reshape(B * Lk, H * D), v.reshape(B * Lk, H * D)], 1).to(dtype).cuda()
        C = H * D
        kd, vd = kv[:, :C], kv[:, C:]
    out = ops.attention(qd, kd, vd, B, H, Lq, Lk, D, scale)
    e, l2 = rel_err(out.float().cpu(), ref)
    assert e < (2e-5 if dtype == torch.float32 else (2e-3 if dtype == torch.float16 else 1.5e-2)), f"{e:.3e} {l2:.3e}"


ATTN_BWD_CASES = [(2, 8, 64, 64, 40), (1, 8, 200, 200, 80), (2, 8, 64, 64, 160), (2, 8, 256, 77, 40),
                  (1, 8, 100, 77, 80), (1, 2, 70, 130, 64), (1, 8, 1024, 1024, 80)]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", ATTN_BWD_CASES, ids=[f"B{c[0]}H{c[1]}q{c[2]}k{c[3]}d{c[4]}" for c in ATTN_BWD_CASES])
def test_attention_backward(cuda, dtype, case):
    """madm_attention_bwd (recomputed log-sum-exp, dq pass + dk/dv pass) equals torch autograd through
    F.scaled_dot_product_attention for self- and cross-attention shapes incl. ragged tails."""
    from madm_amd import ops
    B, H, Lq, Lk, D = case
    q = _q(_gen((B, Lq, H, D), 1), dtype).requires_grad_(True)
    k = _q(_gen((B, Lk, H, D), 2), dtype).requires_grad_(True)
    v = _q(_gen((B, Lk, H, D), 3), dtype).requires_grad_(True)
    scale = D ** -0.5
    ref = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale=scale)
    ref = ref.transpose(1, 2).reshape(B * Lq, H * D)
    do = _q(_gen((B * Lq, H * D), 4), dtype)
    ref.backward(do)
    C = H * D
    qd = q.detach().reshape(B * Lq, C).to(dtype).cuda()
    kv = torch.cat([k.detach().reshape(B * Lk, C), v.detach().reshape(B * Lk, C)], 1).to(dtype).cuda()
    kd, vd = kv[:, :C], kv[:, C:]
    out = ops.attention(qd, kd, vd, B, H, Lq, Lk, D, scale)
    dq, dk, dv = ops.attention_backward(qd, kd, vd, out, do.to(dtype).cuda(), B, H, Lq, Lk, D, scale)
    torch.cuda.synchronize()
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    for name, got, want in (("dq", dq, q.grad), ("dk", dk, k.grad), ("dv", dv, v.grad)):
        rows = B * (Lq if name == "dq" else Lk)
        e, l2 = rel_err(got.float().cpu(), want.reshape(rows, C))
        assert e < tol, f"{name}: {e:.3e} {l2:.3e}"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_vae_attention_gemm_path_equals_flash_path(cuda, dtype):
    """The VAE mid-block attention as GEMMs (S = QK^T in f32, row softmax, P V with V^T from a swapped GEMM,
    bias after P V) against torch SDPA, and the row-softmax kernel on its own."""
    from madm_amd import ops, weights
    from madm_amd.sd_vae import VaeAttention
    from madm_amd.nn import Tok
    C, B, H, W = 512, 2, 16, 16
    att = weights.synth_init_(VaeAttention(C), 3, "a.").cuda()
    x = _q(_gen((B, C, H, W), 1), dtype)
    att.GEMM_MIN_L = 64
    got = att(Tok(to_tokens(x, dtype), B, H, W))
    att.GEMM_MIN_L = 1 << 30
    flash = att(Tok(to_tokens(x, dtype), B, H, W))
    with torch.no_grad():
        sd = {k: v.float().cpu() for k, v in att.state_dict().items()}
        hn = F.group_norm(x, 32, sd["group_norm.weight"], sd["group_norm.bias"], eps=1e-6)
        t = hn.reshape(B, C, H * W).transpose(1, 2)
        q = F.linear(t, sd["to_q.weight"], sd["to_q.bias"])
        k = F.linear(t, sd["to_k.weight"], sd["to_k.bias"])
        v = F.linear(t, sd["to_v.weight"], sd["to_v.bias"])
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        ref = F.linear(o, sd["to_out.0.weight"], sd["to_out.0.bias"]).transpose(1, 2).reshape(B, C, H, W) + x
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(from_tokens(got.t, B, H, W), ref)[0] < tol
    assert rel_err(from_tokens(flash.t, B, H, W), ref)[0] < tol
    s = _gen((37, 256), 9) * 5
    p = ops.softmax_rows(s.cuda(), dtype, 0.3).float().cpu()
    assert rel_err(p, torch.softmax(s * 0.3, -1))[0] < (1e-6 if dtype == torch.float32 else 5e-3)


def test_attention_spike(cuda):
    """Forces the online-softmax rescale: one key dominates late in the sequence."""
    from madm_amd import ops
    B, H, L, D = 1, 8, 192, 40
    q = _gen((B, L, H, D), 1)
    k = _gen((B, L, H, D), 2)
    v = _gen((B, L, H, D), 3)
    k[:, 150] = q[:, 7] * 4.0
    ref = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale=D ** -0.5)
    ref = ref.transpose(1, 2).reshape(B * L, H * D)
    out = ops.attention(q.reshape(B * L, H * D).cuda(), k.reshape(B * L, H * D).cuda(),
                        v.reshape(B * L, H * D).cuda(), B, H, L, L, D, D ** -0.5)
    e, l2 = rel_err(out.cpu(), ref)
    assert e < 2e-5, f"{e:.3e} {l2:.3e}"


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_glue_kernels(cuda, dtype):
    from madm_amd import ops
    kt = ops.k_tile(dtype)
    # image normalisation + layout + range probe
    img = torch.rand((2, 3, 16, 24), generator=torch.Generator().manual_seed(1))
    mm = torch.tensor([float("inf"), float("-inf")], device="cuda")
    t = ops.image_to_nhwc(img.cuda(), dtype, kt, 0.5, 0.5, mm)
    ref = (img - 0.5) / 0.5
    got = from_tokens(t, 2, 16, 24)
    assert rel_err(got[:, :3], ref)[0] < (1e-6 if dtype == torch.float32 else 5e-3)
    assert got[:, 3:].abs().max().item() == 0.0
    mmc = mm.cpu()
    assert abs(mmc[0].item() - ref.min().item()) < 1e-6 and abs(mmc[1].item() - ref.max().item()) < 1e-6
    # im2col rows of the 3x3 stem: a K = k_tile GEMM over them equals the padded 3x3 conv
    cols = ops.image_to_im2col3x3(img.cuda(), dtype, kt, 0.5, 0.5, None).float().cpu()
    unf = F.unfold(ref, 3, padding=1)                       # [B, 27 (c, r, s), HW]
    unf = unf.reshape(2, 3, 9, -1).permute(0, 3, 2, 1).reshape(2 * 16 * 24, 27)   # k = tap*3 + c
    assert rel_err(cols[:, :27], unf)[0] < (1e-6 if dtype == torch.float32 else 5e-3)
    assert cols[:, 27:].abs().max().item() == 0.0
    # nhwc -> nchw
    back = ops.nhwc_to_nchw(t, 2, 3, 16, 24).cpu()
    both = ops.nhwc_to_nchw([t, t], 2, [3, 2], 16, 24).cpu()
    assert torch.equal(both[:, :3], back) and torch.equal(both[:, 3:], back[:, :2])
    assert rel_err(back, got[:, :3])[0] == 0.0
    # timestep embedding
    ts = torch.tensor([0, 60, 999], dtype=torch.int64)
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    arg = ts[:, None].float() * freqs[None]
    ref = torch.cat([torch.cos(arg), torch.sin(arg)], -1)
    emb = ops.timestep_embedding(ts.cuda(), ops.timestep_freqs(320, "cuda"), dtype).float().cpu()
    assert (emb - ref).abs().max().item() < (2e-5 if dtype == torch.float32 else 5e-3)
    # silu / rows_to_f32
    x = _q(_gen((2, 1280), 1), dtype)
    s = ops.silu(x.to(dtype).cuda()).float().cpu()
    assert rel_err(s, F.silu(x))[0] < (1e-6 if dtype == torch.float32 else 5e-3)
    add = _gen((2, 1280), 2)
    r = ops.rows_to_f32(x.to(dtype).cuda(), add.cuda()).cpu()
    assert rel_err(r, x + add)[0] < 1e-6
    # latent scaling + noise mixing
    B, h, w = 2, 8, 8
    mom = _q(_gen((B * h * w, 8), 3), dtype)
    noise = _gen((1, 4, h, w), 4)
    ac = torch.linspace(0.999, 0.01, 1000)
    tsb = torch.tensor([0, 60], dtype=torch.int64)
    lat, noisy = ops.latents_add_noise(mom.to(dtype).cuda(), 0.18215, noise.cuda(), ac.sqrt().cuda(),
                                       (1 - ac).sqrt().cuda(), tsb.cuda(), B, h * w, kt, h, w)
    lat_ref = mom[:, :4].reshape(B, h, w, 4).permute(0, 3, 1, 2) * 0.18215
    assert rel_err(lat.cpu(), lat_ref)[0] < 1e-6
    noisy_ref = ac.sqrt()[tsb][:, None, None, None] * lat_ref + (1 - ac).sqrt()[tsb][:, None, None, None] * noise
    gotn = from_tokens(noisy, B, h, w)
    assert rel_err(gotn[:, :4], noisy_ref)[0] < (1e-6 if dtype == torch.float32 else 5e-3)
    assert gotn[:, 4:].abs().max().item() == 0.0
