"""Tensor-level wrappers over the C ABI (include/madm_hip.h).

PyTorch is used only for device memory, streams and dtype tags: every function hands raw device
pointers of CUDA(HIP) tensors to libmadm_hip.so on the current stream.  Activations are
channels-last 2-D tensors ``[B*H*W, C]``.  There is no fallback path.
"""
import ctypes
import os

import torch

from . import _debug, _lib
from ._lib import (lib, check, Conv2dArgs, AttentionArgs, EPI_NONE, EPI_GEGLU, EPI_RELU, MADM_F32, MADM_BF16, MADM_F16,
                   ACT_NONE, ACT_SILU, ACT_RELU)

_DT = {torch.float32: MADM_F32, torch.bfloat16: MADM_BF16, torch.float16: MADM_F16}
_SUFFIX = {torch.float32: "_f32", torch.bfloat16: "_bf16", torch.float16: "_f16"}


def dtype_code(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise TypeError(f"madm_amd supports float32, bfloat16 and float16 tensors, got {t}")


def k_tile(dtype):
    """K-tile of the MFMA kernels in elements: channel counts must be multiples of it."""
    return 32 if dtype == torch.float32 else 64


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("madm_amd ops need device tensors (the HIP path has no CPU fallback)")


_workspaces = {}

class tuning_profile:
    """``with ops.tuning_profile("latency"):`` -- tile / split-K rows tuned for ONE launch on an idle chip in front of the
    throughput table (madm_set_tuning_profile; process-wide, restored on exit).  ``"throughput"`` (the library default) is what
    the graph runners capture under.  ``pin=True``: contexts opened inside this one WITHOUT pin are ignored -- the runners of
    madm_amd/pipeline.py pin "throughput" around their warm-up and capture, so the ``sync_profile()`` that a plain synchronous
    ``LdmRocm.forward`` / ``MTMADISE.forward_train`` opens (``SYNC_PROFILE``, "latency") does nothing in there."""
    CODES = {"throughput": 0, "latency": 1}
    _pinned = 0

    def __init__(self, name, pin=False):
        self.code, self.pin = self.CODES[name], bool(pin)
        self.active = False

    def __enter__(self):
        if tuning_profile._pinned and not self.pin:
            return self
        self.active = True
        self.prev = lib.madm_get_tuning_profile()
        check(lib.madm_set_tuning_profile(self.code), "madm_set_tuning_profile")
        if self.pin:
            tuning_profile._pinned += 1
        return self

    def __exit__(self, *exc):
        if self.active:
            if self.pin:
                tuning_profile._pinned -= 1
            check(lib.madm_set_tuning_profile(self.prev), "madm_set_tuning_profile")
            self.active = False
        return False


def sync_profile():
    """The profile context of a synchronous (one batch in flight) forward / training step."""
    return tuning_profile(SYNC_PROFILE)


# profile of a plain synchronous forward() call (env MADM_SYNC_PROFILE=throughput for A/B runs and for bit-comparisons with the runners)
SYNC_PROFILE = os.environ.get("MADM_SYNC_PROFILE", "latency")


# ---- lazily built operands and side streams -------------------------------------------------------------------------------
# Packed weights, composed operands and stacked K/V / time-embedding weights are built lazily by the FIRST forward that needs
# them after a parameter change -- on whatever stream that forward runs.  When a pass runs on a side stream beside the main one
# (the EMA teacher of mtmadise.forward_train shares the student's UNet and frozen VAE), an operand it builds there is consumed by
# the main stream's passes a few host statements later with NO ordering between the two streams.  ``side_builds(main)`` is the
# context such a pass runs in; every cache that stores a freshly built operand calls ``note_build()`` right after the build
# kernels are enqueued: inside the context that records an event on the building stream and makes ``main`` wait for it (the
# consumers on main are enqueued later on the host, so they are ordered behind the build).  Outside the context: nothing.
_SIDE_BUILD_MAIN = None
SIDE_BUILDS_NOTED = 0


class side_builds:
    def __init__(self, main_stream):
        self.main = main_stream

    def __enter__(self):
        global _SIDE_BUILD_MAIN
        self.prev, _SIDE_BUILD_MAIN = _SIDE_BUILD_MAIN, self.main
        return self

    def __exit__(self, *exc):
        global _SIDE_BUILD_MAIN
        _SIDE_BUILD_MAIN = self.prev
        return False


def note_build():
    """Call after enqueueing the kernels that fill a cached operand (see ``side_builds``)."""
    global SIDE_BUILDS_NOTED
    main = _SIDE_BUILD_MAIN
    if main is None or torch.cuda.is_current_stream_capturing():
        return
    cur = torch.cuda.current_stream(main.device)
    if cur != main:
        ev = torch.cuda.Event()
        ev.record(cur)
        main.wait_event(ev)
        SIDE_BUILDS_NOTED += 1

# Optional per-launch profiler used by bench.py: when PROFILE is a list, every MFMA-kernel launch
# appends (kernel name, algorithmic FLOPs, start event, end event) recorded on the launch stream.
PROFILE = None
# With PROFILE_DIFF every forward conv / GEMM launch is issued THREE times as [K] [K K] between three events: the
# difference of the two brackets is one launch (incl. the kernel-to-kernel gap) with the constant cost of an event pair
# cancelled -- no overhead estimate to subtract (bench.py's roofline; the pass's results are thrown away: fused
# statistics are accumulated three times).
PROFILE_DIFF = False
_TILE_NAMES = {1: "igemm_128x128", 2: "igemm_128x64", 3: "igemm_64x64", 4: "conv3x3_halo_x128", 5: "conv3x3_halo_x64",
               6: "igemm_64x64d", 7: "igemm_glds_64x64", 8: "igemm_glds_128x64", 9: "conv3x3_halo_dma_x128",
               10: "conv3x3_halo_dma_x64", 11: "igemm_glds_64x64s", 12: "conv3x3_h16_x128", 13: "igemm_apanel",
               14: "igemm_glds_128x128", 15: "igemm_glds_128x128d",
               16: "igemm_glds_64x64d", 17: "igemm_glds_128x64d"}
EXP_NO_STATS = bool(int(__import__('os').environ.get('MADM_EXP_NO_STATS', '0')))   # timing experiment only
FORCE_SPLITK = None   # tools/tune_insitu.py: split-K factor forced on every small-M launch
if os.environ.get("MADM_EXP_SPLITK"):   # experiment: e.g. 1 = no split-K anywhere (does the staged pipeline still want it?)
    FORCE_SPLITK = int(os.environ["MADM_EXP_SPLITK"])
# timing experiments only (results become garbage): launches of the named classes are skipped -- "layernorm", "gn_apply",
# "attention", "softmax", or tile codes of madm_conv2d_pick_tile ("tile7", ...): what would the step cost without them?
EXP_SKIP = set(filter(None, os.environ.get("MADM_EXP_SKIP", "").split(",")))
_SKIP_RE = [__import__("re").compile(t[3:]) for t in EXP_SKIP if t.startswith("re:")]


def _skip_match(desc):
    return any(r.search(desc) for r in _SKIP_RE)


# When TILE_LOG is a list, every forward conv / linear launch appends (description, tile code, split-K, has_tuned_row, FLOPs):
# tests/test_parity_gpu.py::test_bench_workloads_have_tuned_rows
TILE_LOG = None
FUSE_GN = True   # fold GroupNorm(+SiLU) into eligible 3x3 convs (debug switch)
# the GroupNorm that consumes a split-K conv rides on its reduction (conv2d(post_gn=...)); env MADM_NO_POST_GN for A/B runs
POST_GN = not bool(int(os.environ.get("MADM_NO_POST_GN", "0")))
HALO_MIN_W = int(__import__("os").environ.get("MADM_HALO_MIN_W", "8"))   # mirrors halo_min_width() of igemm.hip
import os as _os
FUSE_GN_MAX_N = int(_os.environ.get("MADM_FUSE_GN_MAX_N", "256"))   # ... whose output has at most 256 channels: the
# transform is redone once per output-channel tile, so on the wide UNet layers (320 .. 1280 channels, 5 .. 20 tiles of 64) a
# stand-alone GroupNorm pass + the plain conv is less total work.  Same-box A/B with both variants tuned (images/s,
# overlapped / serial): 128 -> 269 / 220, 256 -> 270 / 220, unlimited -> 262 / 221.  Env override for A/B runs.


def can_fuse_groupnorm(IH, IW, KH, stride, pad, asym_pad, upsample):
    """Mirror of madm_conv2d_can_fuse_groupnorm (the LDS halo-tile 3x3 kernel applies)."""
    return (FUSE_GN and KH == 3 and stride == 1 and pad == 1 and not asym_pad and not upsample and IH >= 8
            and IW >= HALO_MIN_W)


def groupnorm_finalize(stats, B, HW, G, gamma, beta, eps):
    """Channel sums of one or two concatenated sources -> (scale, shift) f32 [B, Ctot] with
    x * scale + shift == GroupNorm(x) (what conv2d(..., gn=...) computes in its own prologue; kept as a utility)."""
    _need_cuda(gamma, beta, *stats)
    C1 = stats[0].shape[1]
    Ctot = sum(st.shape[1] for st in stats)
    assert gamma.numel() == Ctot and all(st.dtype == torch.float64 and st.is_contiguous() for st in stats)
    scale = torch.empty((B, Ctot), dtype=torch.float32, device=gamma.device)
    shift = torch.empty((B, Ctot), dtype=torch.float32, device=gamma.device)
    check(lib.madm_groupnorm_finalize(B, HW, Ctot, G, stats[0].data_ptr(), C1,
                                      stats[1].data_ptr() if len(stats) > 1 else None, gamma.data_ptr(),
                                      beta.data_ptr(), float(eps), scale.data_ptr(), shift.data_ptr(), _stream()),
          "madm_groupnorm_finalize")
    return scale, shift


class _Prof:
    def __init__(self, name, flops, desc="", nbytes=0):
        self.name, self.flops, self.desc, self.nbytes = name, flops, desc, nbytes
        self.e2 = None          # PROFILE_DIFF: recorded after two more launches of the same call

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record()
            PROFILE.append((self.name, self.flops, self.e0, self.e1, self.desc, self.nbytes, self))
        return False


def _workspace(nbytes, device):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return _debug.poison(ws) if _debug.MODE else ws


def conv2d(x1, w, B, IH, IW, *, N, x2=None, KH=1, KW=1, stride=1, pad_t=0, pad_l=0, OH=None, OW=None,
           upsample=False, bias=None, rowvec=None, residual=None, epilogue=EPI_NONE, out=None,
           splitk=None, alg_nk=None, stats=None, gn=None, out_f32=False, ln=None, post_gn=None):
    """Implicit-GEMM conv / linear.  x1: [B*IH*IW, C1] dense; x2 optional second source (concat);
    w: packed [N, KH*KW*(C1+C2)]; returns out [B*OH*OW, N] (N/2 columns for GEGLU).

    ``post_gn`` = (gamma, beta, groups, eps, act): the GroupNorm(+act) that CONSUMES this conv, applied by its split-K
    reduction when the launch has one (madm_conv2d_args.pn_gamma).  Returns (out, applied): ``applied`` False = the launch
    cannot carry it (no split-K, group too large for LDS): ``out`` is then the raw conv output (with ``stats`` filled, if
    given) and the caller runs the norm itself; True = ``out`` holds the normalised activations and ``stats`` was not used."""
    _need_cuda(x1, w, x2, bias, rowvec, residual, out)
    C1 = x1.shape[1]
    C2 = 0 if x2 is None else x2.shape[1]
    assert x1.stride(1) == 1 and x1.shape[0] == B * IH * IW, (x1.shape, B, IH, IW)
    assert x2 is None or (x2.stride(1) == 1 and x2.shape[0] == x1.shape[0] and x2.dtype == x1.dtype)
    assert w.dtype == x1.dtype and w.stride(1) == 1 and tuple(w.shape) == (N, KH * KW * (C1 + C2)), \
        (w.shape, N, KH, KW, C1, C2)
    if OH is None:
        OH = IH * (2 if upsample else 1)
        OW = IW * (2 if upsample else 1)
    M = B * OH * OW
    ocols = N // 2 if epilogue == EPI_GEGLU else N
    odt = torch.float32 if out_f32 else x1.dtype
    if out is None:
        out = torch.empty((M, ocols), dtype=odt, device=x1.device)
    assert out.shape == (M, ocols) and out.stride(1) == 1 and out.dtype == odt
    a = Conv2dArgs()
    a.dtype = dtype_code(x1)
    a.in1 = x1.data_ptr()
    a.in2 = x2.data_ptr() if x2 is not None else None
    a.C1, a.C2 = C1, C2
    a.ld1 = x1.stride(0)
    a.ld2 = x2.stride(0) if x2 is not None else 0
    a.B, a.IH, a.IW, a.OH, a.OW = B, IH, IW, OH, OW
    a.KH, a.KW, a.stride, a.pad_t, a.pad_l = KH, KW, stride, pad_t, pad_l
    a.upsample = 1 if upsample else 0
    a.w = w.data_ptr()
    a.ldw = w.stride(0)
    a.out_f32 = 1 if out_f32 else 0
    a.N = N
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
        a.bias = bias.data_ptr()
    if rowvec is not None:
        assert rowvec.dtype == torch.float32 and rowvec.stride(1) == 1 and tuple(rowvec.shape) == (B, N)
        a.rowvec = rowvec.data_ptr()
        a.ldrv = rowvec.stride(0)
    if residual is not None:
        assert residual.dtype == x1.dtype and residual.shape == (M, ocols) and residual.stride(1) == 1
        a.residual = residual.data_ptr()
        a.ldr = residual.stride(0)
    a.out = out.data_ptr()
    a.ldo = out.stride(0)
    a.epilogue = epilogue
    if gn is not None:   # (sums list, gamma, beta, groups, eps, act): GroupNorm(+act) of the input fused into the conv
        sums, gamma, beta, groups, eps, act = gn
        _need_cuda(gamma, beta, *sums)
        assert len(sums) == (2 if x2 is not None else 1) and gamma.numel() == C1 + C2 and beta.numel() == C1 + C2
        assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.is_contiguous() and beta.is_contiguous()
        for st_, c_ in zip(sums, (C1, C2)):
            assert st_.dtype == torch.float64 and st_.is_contiguous() and st_.numel() == B * c_ * 2
        a.gn_sums1 = sums[0].data_ptr()
        a.gn_sums2 = sums[1].data_ptr() if len(sums) > 1 else None
        a.gn_gamma, a.gn_beta = gamma.data_ptr(), beta.data_ptr()
        a.gn_groups, a.gn_eps, a.gn_act = int(groups), float(eps), _act_code(act=act)
    if ln is not None:   # (colsum f32 [N] of the gamma-scaled packed weight, eps): LayerNorm of the input rows folded in
        cs, eps = ln
        _need_cuda(cs)
        assert cs.dtype == torch.float32 and cs.is_contiguous() and cs.numel() == N and x2 is None and KH == 1 and gn is None
        a.ln_colsum, a.ln_eps = cs.data_ptr(), float(eps)
        splitk = 1       # every workgroup must walk whole rows (the row sums come from its own A fragments)
    a.splitk = 1
    if FORCE_SPLITK is not None and splitk is None:
        nk = KH * KW * (C1 + C2) // k_tile(x1.dtype)
        splitk = FORCE_SPLITK if (M <= 16384 and FORCE_SPLITK <= max(1, nk // 2)) else 1
    if splitk is None:
        splitk = lib.madm_conv2d_suggest_splitk(ctypes.byref(a))
    a.splitk = max(1, int(splitk))
    applied = False
    if post_gn is not None:
        gamma, beta, groups, eps, act = post_gn
        _need_cuda(gamma, beta)
        assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.is_contiguous() and beta.is_contiguous()
        assert gamma.numel() == N and beta.numel() == N and residual is None and ln is None
        a.pn_groups = int(groups)
        if POST_GN and lib.madm_conv2d_can_post_groupnorm(ctypes.byref(a)):
            a.pn_gamma, a.pn_beta, a.pn_eps, a.pn_act = gamma.data_ptr(), beta.data_ptr(), float(eps), _act_code(act=act)
            applied = True
    if stats is not None and not applied:
        assert stats.dtype == torch.float64 and stats.is_contiguous() and stats.numel() == B * N * 2
        a.stats = stats.data_ptr()
    ws = None
    if a.splitk > 1:
        nbytes = lib.madm_conv2d_workspace_bytes(ctypes.byref(a))
        ws = _workspace(nbytes, x1.device)
        a.workspace = ws.data_ptr()
        a.workspace_bytes = ws.numel()
    if EXP_SKIP:
        # timing experiments only (tools/exp/skip_sensitivity.sh): "tileN", or "re:<regex>" on "k<KH> s<stride> M.. N.. K.."
        if ("tile%d" % lib.madm_conv2d_pick_tile(ctypes.byref(a))) in EXP_SKIP or _skip_match(
                f"k{KH} s{stride}{' up' if upsample else ''} M{M} N{N} K{KH * KW * (C1 + C2)}"):
            return out if post_gn is None else (out, applied)
    if TILE_LOG is not None:
        TILE_LOG.append((f"dt{a.dtype} M{M} N{N} K{KH * KW * (C1 + C2)} k{KH} s{stride}{' up' if upsample else ''}"
                         f"{' gn' if gn is not None else ''}", lib.madm_conv2d_pick_tile(ctypes.byref(a)), a.splitk,
                         bool(lib.madm_conv2d_has_tuned_row(ctypes.byref(a))), 2.0 * M * N * KH * KW * (C1 + C2)))
    if PROFILE is None:
        check(lib.madm_conv2d_fwd(ctypes.byref(a), _stream()), "madm_conv2d_fwd")
    else:
        an, ak = alg_nk if alg_nk is not None else (N, KH * KW * (C1 + C2))
        name = _TILE_NAMES[lib.madm_conv2d_pick_tile(ctypes.byref(a))] + _SUFFIX[x1.dtype]
        desc = (f"M{M} N{N} K{KH * KW * (C1 + C2)} k{KH} s{stride}{' up' if upsample else ''}"
                f"{' gn' if gn is not None else ''}{' ln' if ln is not None else ''}"
                f"{' geglu' if epilogue == EPI_GEGLU else ''}{' res' if residual is not None else ''}"
                f" sk{a.splitk}{' +gn' if applied else ''}")
        es = x1.element_size()
        # algorithmic HBM bytes: every input element, weight and output element once
        nbytes = (B * IH * IW * (C1 + C2) * es + w.numel() * es
                  + M * ocols * (4 if out_f32 else es) + (M * ocols * es if residual is not None else 0))
        with _Prof(name, 2.0 * M * an * ak, desc, nbytes) as pr:
            check(lib.madm_conv2d_fwd(ctypes.byref(a), _stream()), "madm_conv2d_fwd")
        if PROFILE_DIFF:
            check(lib.madm_conv2d_fwd(ctypes.byref(a), _stream()), "madm_conv2d_fwd")
            check(lib.madm_conv2d_fwd(ctypes.byref(a), _stream()), "madm_conv2d_fwd")
            pr.e2 = torch.cuda.Event(enable_timing=True)
            pr.e2.record()
    return out if post_gn is None else (out, applied)


def linear(x, w, *, bias=None, residual=None, epilogue=EPI_NONE, out=None, x2=None, splitk=None, alg_nk=None,
           stats=None, B=1, out_f32=False, ln=None):
    """out = x @ w.T (+bias) (+residual); x: [M, K] dense, w: [N, K(+K2)].  ``stats`` ([B, N, 2]) asks for
    the fused GroupNorm statistics of the output, the M rows being B images of M/B tokens."""
    M = x.shape[0]
    assert M % B == 0
    return conv2d(x, w, B, M // B, 1, N=w.shape[0], x2=x2, bias=bias, residual=residual,
                  epilogue=epilogue, out=out, splitk=splitk, alg_nk=alg_nk, stats=stats, out_f32=out_f32, ln=ln)


def conv2d_wgrad(x1, dout, B, IH, IW, *, x2=None, KH=1, KW=1, stride=1, pad_t=0, pad_l=0, OH=None, OW=None,
                 upsample=False, dw=None, splitm=0, dbias=None):
    """Weight gradient of :func:`conv2d` (torch autograd in the reference, engine/train_loop.py:203-217):
    dw[n][(kh, kw, c)] += sum_m dout[m][n] * A(m, k).  x1 / x2: the tensors the forward conv read; dout: [B*OH*OW, N];
    dw: f32 [N, KH*KW*(C1+C2)], ACCUMULATED into (a zeroed one is created when None); dbias: None or f32 [N], accumulated
    into: the column sums of dout (the bias gradient), gathered by the same launch."""
    _need_cuda(x1, x2, dout, dw)
    C1 = x1.shape[1]
    C2 = 0 if x2 is None else x2.shape[1]
    N = dout.shape[1]
    if OH is None:
        OH = IH * (2 if upsample else 1)
        OW = IW * (2 if upsample else 1)
    assert x1.stride(1) == 1 and x1.shape[0] == B * IH * IW, (x1.shape, B, IH, IW)
    assert x2 is None or (x2.stride(1) == 1 and x2.shape[0] == x1.shape[0] and x2.dtype == x1.dtype)
    assert dout.dtype == x1.dtype and dout.stride(1) == 1 and dout.shape[0] == B * OH * OW, (dout.shape, B, OH, OW)
    K = KH * KW * (C1 + C2)
    if dw is None:
        dw = zeros_f32((N, K), x1.device)
    assert dw.dtype == torch.float32 and dw.is_contiguous() and tuple(dw.shape) == (N, K)
    a = _lib.Conv2dWgradArgs()
    a.dtype = dtype_code(x1)
    a.in1 = x1.data_ptr()
    a.in2 = x2.data_ptr() if x2 is not None else None
    a.C1, a.C2 = C1, C2
    a.ld1 = x1.stride(0)
    a.ld2 = x2.stride(0) if x2 is not None else 0
    a.dout, a.ldd = dout.data_ptr(), dout.stride(0)
    a.dw = dw.data_ptr()
    a.B, a.IH, a.IW, a.OH, a.OW = B, IH, IW, OH, OW
    a.KH, a.KW, a.stride, a.pad_t, a.pad_l = KH, KW, stride, pad_t, pad_l
    a.upsample = 1 if upsample else 0
    a.N = N
    a.splitm = int(splitm)
    if dbias is not None:
        _need_cuda(dbias)
        assert dbias.dtype == torch.float32 and dbias.is_contiguous() and dbias.numel() == N
        a.dbias = dbias.data_ptr()
    if PROFILE is None:
        check(lib.madm_conv2d_wgrad(ctypes.byref(a), _stream()), "madm_conv2d_wgrad")
    else:
        es = x1.element_size()
        nbytes = B * IH * IW * (C1 + C2) * es + dout.numel() * es + dw.numel() * 4
        with _Prof("conv2d_wgrad" + _SUFFIX[x1.dtype], 2.0 * dout.shape[0] * N * K,
                   f"M{dout.shape[0]} N{N} K{K} k{KH} s{stride}", nbytes):
            check(lib.madm_conv2d_wgrad(ctypes.byref(a), _stream()), "madm_conv2d_wgrad")
    return dw


_poison_sink = {}


def poison_lds(device, pattern=0x7fc00000):
    """Test aid (madm_debug_poison_lds): every CU's LDS filled with ``pattern`` (default: quiet NaNs) on the current stream."""
    sink = _poison_sink.get(str(device))
    if sink is None:
        sink = _poison_sink[str(device)] = torch.zeros(1, dtype=torch.int32, device=device)
    check(lib.madm_debug_poison_lds(int(pattern), sink.data_ptr(), _stream()), "madm_debug_poison_lds")


def pack_weight(w, dtype, ktile, splits=None, interleave=False):
    """f32 master weight on the GPU ([N, Cin, KH, KW] conv or [N, K] linear) -> the packed forward operand
    [N, KH*KW*sum(pad(splits))] of ``dtype`` in one launch (packing.pack_conv_weight / pack_linear_weight / the row
    interleave of pack_geglu_weight are the torch statements of the same layout)."""
    _need_cuda(w)
    assert w.dtype == torch.float32 and w.dim() in (2, 4)
    w = w.contiguous()
    N, Cin = w.shape[0], w.shape[1]
    taps = 1 if w.dim() == 2 else w.shape[2] * w.shape[3]
    splits = [Cin] if splits is None else [int(c) for c in splits]
    assert sum(splits) == Cin and 1 <= len(splits) <= 4 and taps <= 9
    cols = taps * sum((c + ktile - 1) // ktile * ktile for c in splits)
    out = torch.empty((N, cols), dtype=dtype, device=w.device)
    arr = (ctypes.c_int * len(splits))(*splits)
    check(lib.madm_pack_weight(dtype_code(dtype), w.data_ptr(), out.data_ptr(), cols, N, Cin, taps, len(splits), arr,
                               int(ktile), 1 if interleave else 0, _stream()), "madm_pack_weight")
    return out


def fold_layernorm_pack(w, b, gamma, beta, dtype, interleave=False):
    """(W' = dtype(w * gamma) [N, K], bias' = w beta + b f32 [N], colsum of the rounded W' f32 [N]) in one launch
    (packing.fold_layernorm is the torch statement)."""
    _need_cuda(w, b, gamma, beta)
    assert w.dtype == torch.float32 and w.dim() == 2 and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    w, gamma, beta = w.contiguous(), gamma.contiguous(), beta.contiguous()
    N, K = w.shape
    assert gamma.numel() == K and beta.numel() == K and (b is None or (b.dtype == torch.float32 and b.numel() == N))
    out = torch.empty((N, K), dtype=dtype, device=w.device)
    bias = torch.empty((N,), dtype=torch.float32, device=w.device)
    cs = torch.empty((N,), dtype=torch.float32, device=w.device)
    check(lib.madm_fold_layernorm_pack(dtype_code(dtype), w.data_ptr(), None if b is None else b.contiguous().data_ptr(),
                                       gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), bias.data_ptr(), cs.data_ptr(), N, K,
                                       1 if interleave else 0, _stream()), "madm_fold_layernorm_pack")
    return out, bias, cs


def pack_dgrad_weights(w, taps):
    """w [N, taps*C] (the packed forward layout) -> wt [C, taps*N] with the taps reversed: the weights with which
    :func:`conv2d` applied to dout (pad' = K - 1 - pad) computes the data gradient of a stride-1 conv / linear."""
    _need_cuda(w)
    N, K = w.shape
    assert w.is_contiguous() and K % taps == 0
    C = K // taps
    wt = torch.empty((C, taps * N), dtype=w.dtype, device=w.device)
    check(lib.madm_pack_dgrad_weights(dtype_code(w), w.data_ptr(), wt.data_ptr(), N, taps, C, _stream()),
          "madm_pack_dgrad_weights")
    return wt


def conv2d_dgrad(dout, wt, B, OH, OW, *, C, KH=1, KW=1, pad_t=0, pad_l=0, out=None, residual=None, splitk=None):
    """Data gradient of a STRIDE-1 conv2d / linear whose output has the input's spatial size: the forward kernel on
    dout [B*OH*OW, N] with the repacked weights ``wt = pack_dgrad_weights(w, KH*KW)``; returns din [B*OH*OW, C]
    (+ residual: the gradient that reaches the same tensor through a skip connection)."""
    return conv2d(dout, wt, B, OH, OW, N=C, KH=KH, KW=KW, pad_t=KH - 1 - pad_t, pad_l=KW - 1 - pad_l, OH=OH, OW=OW,
                  out=out, residual=residual, splitk=splitk)


def zero_insert2x(dout, B, OH, OW, H, W):
    """[B*OH*OW, C] -> [B*H*W, C] with dout at the even positions and zeros elsewhere (stride-2 data gradient)."""
    _need_cuda(dout)
    assert dout.is_contiguous() and dout.shape[0] == B * OH * OW
    y = torch.empty((B * H * W, dout.shape[1]), dtype=dout.dtype, device=dout.device)
    check(lib.madm_zero_insert2x(dtype_code(dout), dout.data_ptr(), y.data_ptr(), B, OH, OW, H, W, dout.shape[1],
                                 _stream()), "madm_zero_insert2x")
    return y


def sumpool2x2(x, B, H, W):
    """[B*2H*2W, C] -> [B*H*W, C]: sums of the 2 x 2 blocks (gradient of the nearest-2x upsample)."""
    _need_cuda(x)
    assert x.is_contiguous() and x.shape[0] == B * 4 * H * W
    y = torch.empty((B * H * W, x.shape[1]), dtype=x.dtype, device=x.device)
    check(lib.madm_sumpool2x2(dtype_code(x), x.data_ptr(), y.data_ptr(), B, H, W, x.shape[1], _stream()),
          "madm_sumpool2x2")
    return y


def colsum(x, B, HW, out=None):
    """f32 [B, C] per-image column sums of x [B*HW, C] (row-strided view allowed); accumulated into ``out`` if given."""
    _need_cuda(x, out)
    assert x.stride(1) == 1 and x.shape[0] == B * HW
    if out is None:
        out = zeros_f32((B, x.shape[1]), x.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (B, x.shape[1])
    check(lib.madm_colsum(dtype_code(x), x.data_ptr(), x.stride(0), B, HW, x.shape[1], out.data_ptr(), _stream()),
          "madm_colsum")
    return out


def add(a, b):
    """a + b for two dense tensors of the compute dtype (gradient accumulation)."""
    _need_cuda(a, b)
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and a.dtype == b.dtype
    y = torch.empty_like(a)
    check(lib.madm_add(dtype_code(a), a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream()), "madm_add")
    return y


def silu_backward(x, dy):
    _need_cuda(x, dy)
    assert x.is_contiguous() and dy.is_contiguous() and x.shape == dy.shape and x.dtype == dy.dtype
    dx = torch.empty_like(x)
    check(lib.madm_silu_bwd(dtype_code(x), x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream()),
          "madm_silu_bwd")
    return dx


def softmax_rows(s, dtype, scale):
    """softmax(scale * s) over the last dim of the f32 logits [rows, L] -> [rows, L] of ``dtype``."""
    _need_cuda(s)
    assert s.dtype == torch.float32 and s.stride(1) == 1
    rows, L = s.shape
    out = torch.empty((rows, L), dtype=dtype, device=s.device)
    check(lib.madm_softmax_rows(dtype_code(dtype), s.data_ptr(), out.data_ptr(), rows, L, s.stride(0), out.stride(0),
                                float(scale), _stream()), "madm_softmax_rows")
    return out


class StatsArena:
    """Zero-initialised f64 scratch for the fused GroupNorm statistics: ONE memset per forward instead
    of one per layer.  ``reset()`` at the start of a forward allocates (and zeroes) the size the previous
    forward needed; ``take(n)`` hands out slices and falls back to individual allocations on overflow.
    (``dtype`` f32: the same for the zero-initialised gradient accumulators of one backward pass -- weight-gradient
    tiles, bias / norm parameter sums: 1 400 five-microsecond fills per training step became two memsets.)"""

    def __init__(self, dtype=torch.float64):
        self.dtype = dtype
        self.need = 0
        self.buf = None
        self.off = 0
        self.used = 0

    def reset(self, device):
        self.need = max(self.need, self.used)
        self.buf = torch.zeros(self.need, dtype=self.dtype, device=device) if self.need else None
        self.off = 0
        self.used = 0

    def take(self, n, device):
        n16 = (n + 15) // 16 * 16
        self.used += n16
        if self.buf is not None and self.buf.device == device and self.off + n16 <= self.buf.numel():
            out = self.buf[self.off:self.off + n]
            self.off += n16
            return out
        return torch.zeros(n, dtype=self.dtype, device=device)

    def drop(self):
        """Releases the buffer (slices handed out keep their storage alive)."""
        self.need = max(self.need, self.used)
        self.buf, self.off, self.used = None, 0, 0


ARENA = StatsArena()
GRAD_ZEROS = StatsArena(torch.float32)   # reset per backward pass by the training step (madm_amd/mtmadise.py)


def zeros_f32(shape, device):
    n = 1
    for d in shape:
        n *= int(d)
    return GRAD_ZEROS.take(n, device).view(*shape)


def new_chsums(B, C, device):
    """Zeroed per-(image, channel) statistics buffer [B, C, 2] (f64)."""
    return ARENA.take(B * C * 2, device).view(B, C, 2)


def groupnorm_stats(x, B, HW, chsums):
    """Adds x's per-(image, channel) sum / sum of squares into the zeroed f64 ``chsums`` [B, C, 2]."""
    _need_cuda(x, chsums)
    assert x.is_contiguous() and x.shape[0] == B * HW
    assert chsums.dtype == torch.float64 and chsums.is_contiguous() and chsums.numel() == B * x.shape[1] * 2
    check(lib.madm_groupnorm_stats(dtype_code(x), x.data_ptr(), B, HW, x.shape[1], chsums.data_ptr(), _stream()),
          "madm_groupnorm_stats")


def _act_code(silu=False, act=None):
    if act is None:
        return ACT_SILU if silu else ACT_NONE
    if isinstance(act, str):
        return {"none": ACT_NONE, "silu": ACT_SILU, "relu": ACT_RELU}[act]
    if isinstance(act, bool):
        return ACT_SILU if act else ACT_NONE
    return int(act)


def groupnorm(xs, B, HW, G, gamma, beta, eps, silu=False, stats=None, act=None, residual=None):
    """GroupNorm(+SiLU) of the channel concatenation of one or two sources ``xs`` (a tensor or a list
    of [B*HW, C_i] tensors); ``stats`` = matching list of chsums tensors [B, C_i, 2] (from the producing
    conv's epilogue) or None entries (computed here).  Returns the normalised [B*HW, sum C_i] tensor."""
    if isinstance(xs, torch.Tensor):
        xs = [xs]
    assert 1 <= len(xs) <= 2
    stats = list(stats) if stats is not None else [None] * len(xs)
    for i, x in enumerate(xs):
        if stats[i] is None:
            stats[i] = new_chsums(B, x.shape[1], x.device)
            groupnorm_stats(x, B, HW, stats[i])
    _need_cuda(gamma, beta, *xs)
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
    Ctot = sum(x.shape[1] for x in xs)
    assert gamma.numel() == Ctot
    out = torch.empty((xs[0].shape[0], Ctot), dtype=xs[0].dtype, device=xs[0].device)
    C1 = xs[0].shape[1]
    s2 = stats[1].data_ptr() if len(xs) > 1 else None
    off = 0
    if len(xs) == 2 and residual is None and "gn_apply" not in EXP_SKIP:      # both sources in one launch
        assert xs[0].is_contiguous() and xs[1].is_contiguous()
        check(lib.madm_groupnorm_apply_cat(dtype_code(xs[0]), xs[0].data_ptr(), xs[1].data_ptr(), out.data_ptr(), out.stride(0),
                                           B, HW, C1, xs[1].shape[1], G, stats[0].data_ptr(), s2, gamma.data_ptr(),
                                           beta.data_ptr(), float(eps), _act_code(silu, act), _stream()),
              "madm_groupnorm_apply_cat")
        return out
    for x in xs:
        assert x.is_contiguous()
        if "gn_apply" in EXP_SKIP:
            break
        check(lib.madm_groupnorm_apply(dtype_code(x), x.data_ptr(), out.data_ptr(), out.stride(0), B, HW, x.shape[1],
                                       off, Ctot, G, stats[0].data_ptr(), C1, s2, gamma.data_ptr(), beta.data_ptr(),
                                       float(eps), _act_code(silu, act), _ptr(residual),
                                       residual.stride(0) if residual is not None else 0, _stream()),
              "madm_groupnorm_apply")
        off += x.shape[1]
    return out


def groupnorm_backward(xs, dy, B, HW, G, gamma, beta, eps, stats, act=None, dgamma=None, dbeta=None, dres=None):
    """Backward of :func:`groupnorm` (without residual): ``xs`` / ``stats`` / gamma / beta / eps / act as in the forward
    call, dy [B*HW, Ctot] the gradient of its output.  Returns ([dx_i], dgamma, dbeta); dgamma / dbeta (f32 [Ctot]) are
    accumulated into when given, created zeroed otherwise.  ``dres`` [B*HW, Ctot]: a gradient reaching the same
    (concatenated) input through a skip path, added to the dx_i."""
    if isinstance(xs, torch.Tensor):
        xs = [xs]
    assert 1 <= len(xs) <= 2 and len(stats) == len(xs)
    assert dres is None or (dres.dtype == dy.dtype and dres.stride(1) == 1 and dres.shape == dy.shape and dres.is_cuda)
    _need_cuda(gamma, beta, dy, *xs, *stats)
    Ctot = sum(x.shape[1] for x in xs)
    assert dy.dtype == xs[0].dtype and dy.stride(1) == 1 and tuple(dy.shape) == (B * HW, Ctot)
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == Ctot
    dev = xs[0].device
    if dgamma is None:
        dgamma = zeros_f32((Ctot,), dev)
        dbeta = zeros_f32((Ctot,), dev)
    bsums = ARENA.take(B * Ctot * 2, dev).view(B, Ctot, 2)
    C1 = xs[0].shape[1]
    s2 = stats[1].data_ptr() if len(xs) > 1 else None
    code = _act_code(act=act)
    off = 0
    for x in xs:
        assert x.is_contiguous() and x.shape[0] == B * HW
        check(lib.madm_groupnorm_bwd_sums(dtype_code(x), x.data_ptr(), dy.data_ptr(), dy.stride(0), B, HW, x.shape[1],
                                          off, Ctot, G, stats[0].data_ptr(), C1, s2, gamma.data_ptr(), beta.data_ptr(),
                                          float(eps), code, bsums.data_ptr(), _stream()), "madm_groupnorm_bwd_sums")
        off += x.shape[1]
    dxs, off = [], 0
    for x in xs:
        dx = torch.empty_like(x)
        check(lib.madm_groupnorm_bwd_apply(dtype_code(x), x.data_ptr(), dy.data_ptr(), dy.stride(0), dx.data_ptr(), B, HW,
                                           x.shape[1], off, Ctot, G, stats[0].data_ptr(), C1, s2, gamma.data_ptr(),
                                           beta.data_ptr(), float(eps), code, bsums.data_ptr(), dgamma.data_ptr(),
                                           dbeta.data_ptr(), _ptr(dres), dres.stride(0) if dres is not None else 0,
                                           _stream()), "madm_groupnorm_bwd_apply")
        dxs.append(dx)
        off += x.shape[1]
    return dxs, dgamma, dbeta


def layernorm_backward(x, dy, gamma, eps, dgamma=None, dbeta=None, dres=None):
    """Backward of :func:`layernorm`: returns (dx, dgamma, dbeta); dgamma / dbeta f32 [C] are accumulated into when
    given, created zeroed otherwise; ``dres`` (dense, like x) is added to dx."""
    _need_cuda(x, dy, gamma, dgamma, dbeta, dres)
    assert dres is None or (dres.is_contiguous() and dres.shape == x.shape and dres.dtype == x.dtype)
    assert x.is_contiguous() and dy.is_contiguous() and x.shape == dy.shape and x.dtype == dy.dtype and x.dim() == 2
    if dgamma is None:
        dgamma = zeros_f32((x.shape[1],), x.device)
        dbeta = zeros_f32((x.shape[1],), x.device)
    dx = torch.empty_like(x)
    check(lib.madm_layernorm_bwd(dtype_code(x), x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.shape[0], x.shape[1],
                                 gamma.data_ptr(), float(eps), dgamma.data_ptr(), dbeta.data_ptr(), _ptr(dres), _stream()),
          "madm_layernorm_bwd")
    return dx, dgamma, dbeta


def geglu_backward(pre, dout):
    """pre [M, 2N] interleaved (value, gate) pre-activations, dout [M, N] -> dpre [M, 2N] (same order)."""
    _need_cuda(pre, dout)
    assert pre.is_contiguous() and dout.is_contiguous() and pre.dtype == dout.dtype
    assert pre.shape[0] == dout.shape[0] and pre.shape[1] == 2 * dout.shape[1]
    dpre = torch.empty_like(pre)
    check(lib.madm_geglu_bwd(dtype_code(pre), pre.data_ptr(), dout.data_ptr(), dpre.data_ptr(), pre.shape[0],
                             pre.shape[1], _stream()), "madm_geglu_bwd")
    return dpre


def layernorm(x, gamma, beta, eps, out=None):
    _need_cuda(x, gamma, beta)
    assert x.is_contiguous() and x.dim() == 2
    if out is None:
        out = torch.empty_like(x)
    if "layernorm" in EXP_SKIP:
        return out
    check(lib.madm_layernorm_fwd(dtype_code(x), x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                 gamma.data_ptr(), beta.data_ptr(), float(eps), _stream()),
          "madm_layernorm_fwd")
    return out


def attention(q, k, v, B, H, Lq, Lk, D, scale, out=None):
    """q: [B*Lq, >=H*D] view with unit column stride (row stride = ld), k/v: [B*Lk, ...] views.
    Returns o [B*Lq, H*D]."""
    _need_cuda(q, k, v)
    for t in (q, k, v):
        assert t.stride(1) == 1 and t.dtype == q.dtype
    if out is None:
        out = torch.empty((B * Lq, H * D), dtype=q.dtype, device=q.device)
    a = AttentionArgs()
    a.dtype = dtype_code(q)
    a.q, a.k, a.v, a.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.B, a.H, a.Lq, a.Lk, a.D = B, H, Lq, Lk, D
    a.scale = float(scale)
    if "attention" in EXP_SKIP or (EXP_SKIP and _skip_match(f"attn d{D} Lq{Lq} Lk{Lk}")):
        return out
    with _Prof(f"attn_d{D}" + _SUFFIX[q.dtype], 4.0 * B * H * Lq * Lk * D,
               f"B{B} H{H} Lq{Lq} Lk{Lk}") as pr:
        check(lib.madm_attention_fwd(ctypes.byref(a), _stream()), "madm_attention_fwd")
    if PROFILE is not None and PROFILE_DIFF:   # [K] [K K] like the conv / GEMM launches (the launch is idempotent)
        check(lib.madm_attention_fwd(ctypes.byref(a), _stream()), "madm_attention_fwd")
        check(lib.madm_attention_fwd(ctypes.byref(a), _stream()), "madm_attention_fwd")
        pr.e2 = torch.cuda.Event(enable_timing=True)
        pr.e2.record()
    return out


def attention_backward(q, k, v, o, dout, B, H, Lq, Lk, D, scale, outs=None):
    """Backward of :func:`attention`: q / k / v / o / dout as in the forward (row-strided views with unit column
    stride); returns (dq [B*Lq, H*D], dk [B*Lk, H*D], dv [B*Lk, H*D]) -- dense, or the row-strided views given
    as ``outs`` (e.g. the three column windows of one [M, 3C] buffer for the fused QKV projection's backward)."""
    _need_cuda(q, k, v, o, dout)
    for t in (q, k, v, o, dout):
        assert t.stride(1) == 1 and t.dtype == q.dtype
    if outs is not None:
        dq, dk, dv = outs
        for t, rows in ((dq, B * Lq), (dk, B * Lk), (dv, B * Lk)):
            assert t.is_cuda and t.stride(1) == 1 and t.dtype == q.dtype and tuple(t.shape) == (rows, H * D)
    else:
        dq = torch.empty((B * Lq, H * D), dtype=q.dtype, device=q.device)
        dk = torch.empty((B * Lk, H * D), dtype=q.dtype, device=q.device)
        dv = torch.empty((B * Lk, H * D), dtype=q.dtype, device=q.device)
    a = _lib.AttentionBwdArgs()
    a.dtype = dtype_code(q)
    a.q, a.k, a.v, a.o, a.dout = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), dout.data_ptr()
    a.dq, a.dk, a.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    a.ldq, a.ldk, a.ldv, a.ldo, a.lddo = q.stride(0), k.stride(0), v.stride(0), o.stride(0), dout.stride(0)
    a.lddq, a.lddk, a.lddv = dq.stride(0), dk.stride(0), dv.stride(0)
    a.B, a.H, a.Lq, a.Lk, a.D = B, H, Lq, Lk, D
    a.scale = float(scale)
    nbytes = lib.madm_attention_bwd_workspace_bytes(ctypes.byref(a))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    with _Prof(f"attn_bwd_d{D}" + _SUFFIX[q.dtype], 14.0 * B * H * Lq * Lk * D,
               f"B{B} H{H} Lq{Lq} Lk{Lk}"):
        check(lib.madm_attention_bwd(ctypes.byref(a), _stream()), "madm_attention_bwd")
    return dq, dk, dv


def image_to_nhwc(img, dtype, Cpad, mean, std, minmax=None):
    _need_cuda(img, minmax)
    assert img.dtype == torch.float32 and img.is_contiguous() and img.dim() == 4
    B, C, H, W = img.shape
    out = torch.empty((B * H * W, Cpad), dtype=dtype, device=img.device)
    check(lib.madm_image_to_nhwc(dtype_code(dtype), img.data_ptr(), out.data_ptr(), B, C, H, W, Cpad,
                                 float(mean), float(std), _ptr(minmax), _stream()), "madm_image_to_nhwc")
    return out


def stem_conv3x3(img, wT, bias, dtype, mean, std, stats=None, minmax=None):
    """The SD VAE stem conv straight from the f32 NCHW image (madm_stem_conv3x3): [B,3,H,W] -> tokens [B*H*W, 128]."""
    _need_cuda(img, wT, bias, stats, minmax)
    assert img.dtype == torch.float32 and img.is_contiguous() and img.dim() == 4 and img.shape[1] == 3
    assert wT.dtype == torch.float32 and wT.is_contiguous() and wT.shape[0] == 27 and bias.dtype == torch.float32
    B, _, H, W = img.shape
    N = wT.shape[1]
    out = torch.empty((B * H * W, N), dtype=dtype, device=img.device)
    with _Prof("stem_conv3x3" + _SUFFIX[dtype], 2.0 * B * H * W * N * 27, f"B{B} {H}x{W} N{N}",
               img.numel() * 4 + out.numel() * out.element_size()):
        check(lib.madm_stem_conv3x3(dtype_code(dtype), img.data_ptr(), wT.data_ptr(), bias.data_ptr(), out.data_ptr(), N, B, H,
                                    W, N, float(mean), float(std), _ptr(stats), _ptr(minmax), _stream()),
              "madm_stem_conv3x3")
    return out


def image_to_im2col3x3(img, dtype, Kpad, mean, std, minmax=None):
    """[B,3,H,W] f32 -> im2col rows [B*H*W, Kpad] of the normalised image for a 3x3/pad-1 stem conv."""
    _need_cuda(img, minmax)
    assert img.dtype == torch.float32 and img.is_contiguous() and img.dim() == 4 and img.shape[1] == 3
    B, C, H, W = img.shape
    out = torch.empty((B * H * W, Kpad), dtype=dtype, device=img.device)
    check(lib.madm_image_to_im2col3x3(dtype_code(dtype), img.data_ptr(), out.data_ptr(), B, H, W, Kpad,
                                      float(mean), float(std), _ptr(minmax), _stream()), "madm_image_to_im2col3x3")
    return out


def nchw_to_nhwc(x, dtype, Cpad):
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    B, C, H, W = x.shape
    out = torch.empty((B * H * W, Cpad), dtype=dtype, device=x.device)
    check(lib.madm_nchw_f32_to_nhwc(dtype_code(dtype), x.data_ptr(), out.data_ptr(), B, C, H * W, Cpad,
                                    _stream()), "madm_nchw_f32_to_nhwc")
    return out


def nhwc_to_nchw(xs, B, C, H, W):
    """xs: tensor or list of [B*H*W, ld_i] tensors whose leading channels (C, or a list of counts)
    are concatenated into one f32 [B, sum C_i, H, W] tensor."""
    if isinstance(xs, torch.Tensor):
        xs, C = [xs], [C]
    _need_cuda(*xs)
    Ctot = sum(C)
    out = torch.empty((B, Ctot, H, W), dtype=torch.float32, device=xs[0].device)
    off = 0
    for x, c in zip(xs, C):
        assert x.stride(1) == 1 and x.shape[1] >= c
        check(lib.madm_nhwc_to_nchw_f32(dtype_code(x), x.data_ptr(), x.stride(0), out.data_ptr(), B, c, off, Ctot,
                                        H * W, _stream()), "madm_nhwc_to_nchw_f32")
        off += c
    return out


def copy_columns(src, dst, C):
    """dst[:, :C] = src[:, :C] for 2-D row-strided tensors of the same dtype."""
    _need_cuda(src, dst)
    assert src.dtype == dst.dtype and src.stride(1) == 1 and dst.stride(1) == 1 and src.shape[0] == dst.shape[0]
    check(lib.madm_copy_columns(dtype_code(src), src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0),
                                src.shape[0], C, _stream()), "madm_copy_columns")
    return dst


def clamp_f32(x, lo, hi):
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty_like(x)
    check(lib.madm_clamp_f32(x.data_ptr(), out.data_ptr(), x.numel(), float(lo), float(hi), _stream()),
          "madm_clamp_f32")
    return out


def cast_from_f32(x, dtype):
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib.madm_cast_from_f32(dtype_code(dtype), x.data_ptr(), out.data_ptr(), x.numel(), _stream()),
          "madm_cast_from_f32")
    return out


def latents_add_noise(moments, scaling, noise, sqrt_ac, sqrt_1mac, timesteps, B, HW, Cpad, h, w):
    _need_cuda(moments, noise, sqrt_ac, sqrt_1mac, timesteps)
    assert timesteps.dtype == torch.int64 and noise.dtype == torch.float32 and noise.numel() == 4 * HW
    latents = torch.empty((B, 4, h, w), dtype=torch.float32, device=moments.device)
    noisy = torch.empty((B * HW, Cpad), dtype=moments.dtype, device=moments.device)
    check(lib.madm_latents_add_noise(dtype_code(moments), moments.data_ptr(), moments.stride(0), float(scaling),
                                     noise.data_ptr(), sqrt_ac.data_ptr(), sqrt_1mac.data_ptr(),
                                     timesteps.data_ptr(), latents.data_ptr(), noisy.data_ptr(), B, HW, Cpad,
                                     _stream()), "madm_latents_add_noise")
    return latents, noisy


def timestep_freqs(dim, device):
    """Constant table f_i = exp(-ln(10000) * i / (dim/2)) of diffusers' Timesteps (host-built once)."""
    import math
    half = dim // 2
    return torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half).to(device)


def timestep_embedding(timesteps, freqs, dtype):
    _need_cuda(timesteps, freqs)
    assert timesteps.dtype == torch.int64 and freqs.dtype == torch.float32
    B = timesteps.numel()
    dim = 2 * freqs.numel()
    out = torch.empty((B, dim), dtype=dtype, device=timesteps.device)
    check(lib.madm_timestep_embedding(dtype_code(dtype), timesteps.data_ptr(), freqs.data_ptr(), out.data_ptr(),
                                      B, dim, _stream()), "madm_timestep_embedding")
    return out


def silu(x, out=None):
    _need_cuda(x)
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(lib.madm_silu(dtype_code(x), x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "madm_silu")
    return out


def rows_to_f32(x, add=None):
    _need_cuda(x, add)
    assert x.is_contiguous() and (add is None or (add.dtype == torch.float32 and add.is_contiguous()
                                                   and add.numel() == x.numel()))
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib.madm_rows_to_f32(dtype_code(x), x.data_ptr(), _ptr(add), out.data_ptr(), x.numel(), _stream()),
          "madm_rows_to_f32")
    return out


def resize_bilinear(x, B, IH, IW, OH, OW, out=None):
    """F.interpolate(bilinear, align_corners=False) on tokens x [B*IH*IW, C] -> [B*OH*OW, C]; ``out`` may be a
    column window of a wider concatenation buffer."""
    _need_cuda(x, out)
    assert x.stride(1) == 1 and x.shape[0] == B * IH * IW
    C = x.shape[1]
    if out is None:
        out = torch.empty((B * OH * OW, C), dtype=x.dtype, device=x.device)
    assert out.shape == (B * OH * OW, C) and out.stride(1) == 1 and out.dtype == x.dtype
    check(lib.madm_resize_bilinear(dtype_code(x), x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), B, IH, IW, OH, OW,
                                   C, _stream()), "madm_resize_bilinear")
    return out


def resize_bilinear_nchw(x, OH, OW):
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    B, C, IH, IW = x.shape
    out = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
    check(lib.madm_resize_bilinear_nchw_f32(x.data_ptr(), out.data_ptr(), B * C, IH, IW, OH, OW, _stream()),
          "madm_resize_bilinear_nchw_f32")
    return out


def dwconv3x3(x, w9c, scale, shift, B, H, W, dilation, act=ACT_RELU, out=None):
    """Depthwise dilated 3x3 conv + per-channel affine + activation on tokens x [B*H*W, C]; w9c f32 [9, C]."""
    _need_cuda(x, w9c, scale, shift, out)
    C = x.shape[1]
    assert x.is_contiguous() and x.shape[0] == B * H * W and tuple(w9c.shape) == (9, C) and w9c.dtype == torch.float32
    if out is None:
        out = torch.empty_like(x)
    assert out.stride(1) == 1 and out.shape == x.shape
    check(lib.madm_dwconv3x3(dtype_code(x), x.data_ptr(), w9c.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
                             out.stride(0), B, H, W, C, dilation, act, _stream()), "madm_dwconv3x3")
    return out


def tanh_gate(x1, a1=None, x2=None, a2=None, repeat=1):
    """repeat x [tanh(a1) * x1 + tanh(a2) * x2] (f32, flattened): returns [repeat, *x1.shape[1:]] when x1 has a
    leading 1-dim, else [repeat, *x1.shape]."""
    _need_cuda(x1, a1, x2, a2)
    for t in (x1, a1, x2, a2):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == x1.numel())
    tail = tuple(x1.shape[1:]) if x1.dim() > 1 and x1.shape[0] == 1 else tuple(x1.shape)
    out = torch.empty((repeat,) + tail, dtype=torch.float32, device=x1.device)
    check(lib.madm_tanh_gate(_ptr(a1), x1.data_ptr(), _ptr(a2), _ptr(x2), out.data_ptr(), x1.numel(), repeat, _stream()),
          "madm_tanh_gate")
    return out


def argmax_nchw(x):
    """[B, K, H, W] f32 logits -> int64 [B, H, W], first maximal channel (torch.argmax(dim=1))."""
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    B, K, H, W = x.shape
    out = torch.empty((B, H, W), dtype=torch.int64, device=x.device)
    check(lib.madm_argmax_nchw_f32(x.data_ptr(), out.data_ptr(), B, K, H * W, _stream()), "madm_argmax_nchw_f32")
    return out


def scale_pad_nchw(x, scale, OH, OW, y1=0, x1=0, out=None):
    """x[:, :, y1:y1+OH, x1:x1+OW] * scale with zeros outside x (f32 NCHW): padding, cropping, window extraction."""
    _need_cuda(x, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    B, C, IH, IW = x.shape
    if out is None:
        out = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and tuple(out.shape) == (B, C, OH, OW) and out.dtype == torch.float32
    check(lib.madm_scale_pad_crop_nchw_f32(x.data_ptr(), out.data_ptr(), B * C, IH, IW, y1, x1, OH, OW, float(scale),
                                           _stream()), "madm_scale_pad_crop_nchw_f32")
    return out


def crop_nchw(x, OH, OW, y1=0, x1=0):
    """x[:, :, y1:y1+OH, x1:x1+OW] as a dense tensor."""
    return scale_pad_nchw(x, 1.0, OH, OW, y1, x1)


def slide_merge(win, nW, B, h, w, Wc, x1):
    """Averages window feature maps win [nW*B*h*w, C] (window-major) into the canvas [B*h*Wc, C]."""
    _need_cuda(win)
    assert win.is_contiguous() and win.shape[0] == nW * B * h * w and len(x1) == nW
    out = torch.empty((B * h * Wc, win.shape[1]), dtype=win.dtype, device=win.device)
    xs = (ctypes.c_int * nW)(*[int(v) for v in x1])
    check(lib.madm_slide_merge(dtype_code(win), win.data_ptr(), out.data_ptr(), nW, B, h, w, Wc, win.shape[1], xs, _stream()),
          "madm_slide_merge")
    return out


def confusion_matrix(pred, gt, num_classes, ignore_label, conf=None):
    """conf[(K+1) * pred + gt'] += 1 (int64 [(K+1), (K+1)]), gt' = K where gt == ignore_label."""
    _need_cuda(pred, gt, conf)
    assert pred.dtype == torch.int64 and gt.dtype == torch.int64 and pred.numel() == gt.numel()
    K = num_classes
    if conf is None:
        conf = torch.zeros((K + 1, K + 1), dtype=torch.int64, device=pred.device)
    check(lib.madm_confusion_matrix(pred.contiguous().data_ptr(), gt.contiguous().data_ptr(), pred.numel(), K,
                                    int(ignore_label), conf.data_ptr(), _stream()), "madm_confusion_matrix")
    return conf


# ----------------------------------------------------------------------------- training step (csrc/train.hip)
def scale_channels(x, scale, B, HW, out=None):
    """x [B*HW, C] (row-strided allowed) * scale f32 [B, C] per (image, channel): Dropout2d and its backward."""
    _need_cuda(x, scale, out)
    C = x.shape[1]
    assert x.stride(1) == 1 and x.shape[0] == B * HW and scale.dtype == torch.float32 and scale.is_contiguous() \
        and tuple(scale.shape) == (B, C)
    if out is None:
        out = torch.empty((x.shape[0], C), dtype=x.dtype, device=x.device)
    assert out.stride(1) == 1 and out.shape == x.shape and out.dtype == x.dtype
    check(lib.madm_scale_channels(dtype_code(x), x.data_ptr(), x.stride(0), scale.data_ptr(), out.data_ptr(), out.stride(0),
                                  B, HW, C, _stream()), "madm_scale_channels")
    return out


def relu_backward(y, dy):
    """dy where the ReLU OUTPUT y is positive, else 0 (dense tensors of the compute dtype)."""
    _need_cuda(y, dy)
    assert y.is_contiguous() and dy.is_contiguous() and y.shape == dy.shape and y.dtype == dy.dtype
    dx = torch.empty_like(y)
    check(lib.madm_relu_bwd(dtype_code(y), y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), _stream()), "madm_relu_bwd")
    return dx


def dwconv3x3_wgrad(x, dy, B, H, W, dilation, dw=None):
    """f32 [9, C] weight gradient of :func:`dwconv3x3` (before its affine), accumulated into ``dw`` when given."""
    _need_cuda(x, dy, dw)
    C = x.shape[1]
    assert x.is_contiguous() and x.shape[0] == B * H * W and dy.stride(1) == 1 and dy.shape == x.shape and dy.dtype == x.dtype
    if dw is None:
        dw = zeros_f32((9, C), x.device)
    assert dw.dtype == torch.float32 and dw.is_contiguous() and tuple(dw.shape) == (9, C)
    check(lib.madm_dwconv3x3_wgrad(dtype_code(x), x.data_ptr(), dy.data_ptr(), dy.stride(0), dw.data_ptr(), B, H, W, C,
                                   int(dilation), _stream()), "madm_dwconv3x3_wgrad")
    return dw


def resize_bilinear_backward(dout, B, IH, IW, OH, OW):
    """Adjoint of :func:`resize_bilinear`: dout [B*OH*OW, C] (row-strided allowed) -> din [B*IH*IW, C]."""
    _need_cuda(dout)
    C = dout.shape[1]
    assert dout.stride(1) == 1 and dout.shape[0] == B * OH * OW
    din = torch.empty((B * IH * IW, C), dtype=dout.dtype, device=dout.device)
    nbytes = lib.madm_resize_bilinear_bwd_workspace_bytes(B, IW, OH, C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dout.device)
    check(lib.madm_resize_bilinear_bwd(dtype_code(dout), dout.data_ptr(), dout.stride(0), din.data_ptr(), B, IH, IW, OH, OW, C,
                                       ws.data_ptr(), nbytes, _stream()), "madm_resize_bilinear_bwd")
    return din


def softmax_ce(logits, K, labels, weight=None, ignore_index=255, loss_sum=None, gscale=None, coef=1.0, grad_dtype=None,
               ldd=None):
    """Pixel-weighted cross entropy on f32 logit tokens [M, >=K]: adds sum_i w_i * CE_i into ``loss_sum`` (f64 [1]) and /
    or returns dlogits [M, ldd] of ``grad_dtype`` = coef * gscale * w_i * (softmax - onehot).  Returns (loss_sum, dlogits)."""
    _need_cuda(logits, labels, weight, loss_sum, gscale)
    M = logits.shape[0]
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and logits.shape[1] >= K
    labels = labels.reshape(-1)
    assert labels.dtype == torch.int64 and labels.is_contiguous() and labels.numel() == M
    if weight is not None:
        weight = weight.reshape(-1)
        assert weight.dtype == torch.float32 and weight.is_contiguous() and weight.numel() == M
    assert loss_sum is None or (loss_sum.dtype == torch.float64 and loss_sum.numel() == 1)
    assert gscale is None or (gscale.dtype == torch.float32 and gscale.numel() == 1)
    dl = None
    dt = MADM_F32
    if grad_dtype is not None:
        ldd = k_tile(grad_dtype) if ldd is None else ldd
        dl = torch.empty((M, ldd), dtype=grad_dtype, device=logits.device)
        dt = dtype_code(grad_dtype)
    check(lib.madm_softmax_ce(dt, logits.data_ptr(), logits.stride(0), K, labels.data_ptr(), _ptr(weight), int(ignore_index),
                              M, _ptr(loss_sum), _ptr(gscale), float(coef), _ptr(dl), 0 if dl is None else dl.stride(0),
                              _stream()), "madm_softmax_ce")
    return loss_sum, dl


def masked_l1(pred, gt, mask=None, l2=False, loss_sum=None, gscale=None, coef=1.0, want_grad=False):
    """sum |pred - gt| * nearest(mask) (or the squared difference) over NCHW f32 tensors into ``loss_sum`` (f64 [1]) and /
    or the gradient w.r.t. pred (coef * gscale * mask * sign / 2 d).  mask: f32 [B, 1, Hm, Wm] or None."""
    _need_cuda(pred, gt, mask, loss_sum, gscale)
    B, C, h, w = pred.shape
    assert pred.dtype == torch.float32 and pred.is_contiguous() and gt.dtype == torch.float32 and gt.is_contiguous() \
        and gt.shape == pred.shape
    Hm = Wm = 0
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.is_contiguous() and mask.shape[0] == B and mask.numel() == B * mask.shape[-2] * mask.shape[-1]
        Hm, Wm = mask.shape[-2:]
    dp = torch.empty_like(pred) if want_grad else None
    check(lib.madm_masked_l1(pred.data_ptr(), gt.data_ptr(), _ptr(mask), B, C, h, w, Hm, Wm, 1 if l2 else 0, _ptr(loss_sum),
                             _ptr(gscale), float(coef), _ptr(dp), _stream()), "madm_masked_l1")
    return loss_sum, dp


def tanh_gate_backward(dout, x1, a1=None, x2=None, a2=None, da1=None, dx1=None, da2=None, dx2=None):
    """Backward of :func:`tanh_gate` for dout [repeat, n]: ACCUMULATES into the given f32 gradient buffers."""
    _need_cuda(dout, x1, a1, x2, a2, da1, dx1, da2, dx2)
    n = x1.numel()
    assert dout.dtype == torch.float32 and dout.is_contiguous() and dout.numel() % n == 0
    for t in (x1, a1, x2, a2, da1, dx1, da2, dx2):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n)
    check(lib.madm_tanh_gate_bwd(_ptr(a1), x1.data_ptr(), _ptr(a2), _ptr(x2), dout.data_ptr(), _ptr(da1), _ptr(dx1), _ptr(da2),
                                 _ptr(dx2), n, dout.numel() // n, _stream()), "madm_tanh_gate_bwd")


def batch_stats(x, stats=None, running=None, momentum=0.1):
    """f64 [1, C, 2] sum / sum of squares of x [M, C] over ALL rows -- the statistics of a train-mode BatchNorm2d over
    channels-last tokens.  ``stats``: per-image sums [B, C, 2] from a producing conv's epilogue (computed here when None);
    ``running`` = (running_mean, running_var) f32 [C] buffers updated like nn.BatchNorm2d does in train mode."""
    M, C = x.shape
    if stats is None:
        stats = torch.zeros((1, C, 2), dtype=torch.float64, device=x.device)
        groupnorm_stats(x, 1, M, stats)
    assert stats.dtype == torch.float64 and stats.is_contiguous() and stats.shape[1] == C
    st = torch.empty((1, C, 2), dtype=torch.float64, device=x.device)
    rm, rv = running if running is not None else (None, None)
    _need_cuda(stats, rm, rv)
    assert rm is None or (rm.dtype == torch.float32 and rm.is_contiguous() and rv.dtype == torch.float32 and rv.is_contiguous())
    check(lib.madm_bn_fold_stats(stats.data_ptr(), stats.shape[0], C, float(M), float(momentum), st.data_ptr(), _ptr(rm),
                                 _ptr(rv), _stream()), "madm_bn_fold_stats")
    return st


def batchnorm_train(x, st, gamma, beta, eps, act=None, out=None):
    """Train-mode BatchNorm2d (+act) on tokens x [M, C] with batch statistics ``st`` (:func:`batch_stats`): GroupNorm with
    one image of M pixels and one group per channel (biased variance, like nn.BatchNorm2d's normalisation).  ``out`` may be
    a column window of a wider buffer."""
    _need_cuda(x, st, gamma, beta, out)
    M, C = x.shape
    assert x.is_contiguous() and st.dtype == torch.float64 and st.is_contiguous() and st.numel() == 2 * C
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C
    if out is None:
        out = torch.empty_like(x)
    assert out.stride(1) == 1 and out.shape == x.shape and out.dtype == x.dtype
    check(lib.madm_groupnorm_apply(dtype_code(x), x.data_ptr(), out.data_ptr(), out.stride(0), 1, M, C, 0, C, C,
                                   st.data_ptr(), C, None, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                   _act_code(act=act), None, 0, _stream()), "madm_groupnorm_apply")
    return out


def batchnorm_backward(x, st, dy, gamma, beta, eps, act=None):
    """Backward of :func:`batchnorm_train`: (dx, dgamma, dbeta); dy may be a row-strided column window."""
    M, C = x.shape
    (dx,), dg, db = groupnorm_backward([x], dy, 1, M, C, gamma, beta, eps, [st], act=act)
    return dx, dg, db
