"""Leaf modules of the HIP path: parameter containers with diffusers-compatible names whose
``forward`` launches libmadm_hip kernels on channels-last token tensors.

fp32 master parameters live in ordinary ``nn.Parameter``s (so ``state_dict()`` / checkpoints /
``deepcopy`` for the EMA teacher behave as in the reference); the packed [N][K]
device operands in the compute dtype are derived lazily and re-derived when a parameter's version
counter moves (optimizer steps, ``load_state_dict``).
"""
import torch
import torch.nn as nn

from . import ops, packing


class Tok:
    """Channels-last activation: ``t`` is [B*H*W, C] (contiguous), plus its image geometry."""
    __slots__ = ("t", "B", "H", "W", "stats")

    def __init__(self, t, B, H, W, stats=None):
        assert t.dim() == 2 and t.shape[0] == B * H * W, (t.shape, B, H, W)
        self.t, self.B, self.H, self.W = t, B, H, W
        # f64 [B, C, 2] per-(image, channel) sum / sum of squares written by the producing kernel's
        # epilogue (None: a consuming GroupNorm computes them itself)
        self.stats = stats

    @property
    def C(self):
        return self.t.shape[1]

    @property
    def HW(self):
        return self.H * self.W

    def like(self, t, H=None, W=None):
        return Tok(t, self.B, self.H if H is None else H, self.W if W is None else W)

    def nchw(self, C=None):
        """f32 NCHW copy of the first C channels (hand-over to torch-side consumers)."""
        return ops.nhwc_to_nchw(self.t, self.B, self.C if C is None else C, self.H, self.W)


class _Packed(nn.Module):
    """Cache of packed operands keyed by (dtype, variant), invalidated by parameter versions."""

    def _versions(self):
        return tuple(p._version for p in self.parameters(recurse=False)) + \
            tuple(p.data_ptr() for p in self.parameters(recurse=False))

    def _cache_get(self, key, build, ver=None):
        """``ver``: the version tuple this entry depends on (default: every own parameter, ``_versions()``)."""
        cache = self.__dict__.setdefault("_pack_cache", {})
        ver = self._versions() if ver is None else ver
        hit = cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        with torch.no_grad():
            val = build()
        ops.note_build()       # built on a side stream beside the main one: main's consumers wait for it (ops.side_builds)
        cache[key] = (ver, val)
        return val

    def __deepcopy__(self, memo):
        # packed device operands are derived data: drop them from copies (EMA teacher, cmdise.py:307-335)
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_pack_cache":
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new


class Conv2d(_Packed):
    """nn.Conv2d parameters (weight [N, Cin, k, k], bias [N]); forward = madm_conv2d_fwd.

    ``asym_pad``: diffusers Downsample2D(padding=0): F.pad(x, (0,1,0,1)) then stride-2 conv."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, asym_pad=False, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.asym_pad = kernel_size, stride, padding, asym_pad
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None

    @property
    def n_pad(self):
        return packing.round_up(self.out_channels, 4)

    def packed(self, dtype, splits=None):
        key = (dtype, tuple(splits) if splits else None)

        def build():
            w = self.weight.detach().float()
            dev = w.device
            npad = self.n_pad
            if npad != self.out_channels:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, npad - self.out_channels))
            wp = packing.pack_conv_weight(w, dtype, ops.k_tile(dtype), splits)
            b = None
            if self.bias is not None:
                b = self.bias.detach().float()
                if npad != self.out_channels:
                    b = torch.nn.functional.pad(b, (0, npad - self.out_channels))
                b = b.contiguous()
            return wp.to(dev), b

        return self._cache_get(key, build)

    def out_hw(self, H, W, upsample=False):
        if upsample:
            H, W = 2 * H, 2 * W
        k, s = self.kernel_size, self.stride
        if self.asym_pad:
            return (H + 1 - k) // s + 1, (W + 1 - k) // s + 1
        p = self.padding
        return (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1

    def forward(self, x, x2=None, upsample=False, rowvec=None, residual=None, out=None, stats=True, norm=None,
                act=True, post_norm=None):
        """x (and optional x2 = channel-concatenated second source) are Tok; returns Tok.  ``stats``:
        also emit the GroupNorm statistics of the output from the epilogue (Tok.stats).  ``norm``: a
        GroupNorm module applied (with SiLU when ``act``) to [x | x2] before the conv -- folded into the
        conv's LDS halo load when the 3x3 halo kernel applies, a separate pass otherwise.  ``post_norm``: a GroupNorm
        module applied (with SiLU) to the OUTPUT: the returned Tok holds silu(post_norm(conv(x))) -- by the conv's split-K
        reduction when it has one (ops.conv2d post_gn), by a separate pass otherwise."""
        dtype = x.t.dtype
        gn = None
        if norm is not None:
            # folding the norm into the conv re-applies it once per output-channel tile: measured on MI355X it pays up to
            # 256 output channels (ops.FUSE_GN_MAX_N); wider layers run gn_apply once + the plain conv
            if self.out_channels <= ops.FUSE_GN_MAX_N and ops.can_fuse_groupnorm(
                    x.H, x.W, self.kernel_size, self.stride, self.padding, self.asym_pad, upsample):
                srcs = [x] if x2 is None else [x, x2]
                sts = []
                for s_ in srcs:
                    if s_.stats is None:
                        s_.stats = ops.new_chsums(s_.B, s_.C, s_.t.device)
                        ops.groupnorm_stats(s_.t, s_.B, s_.HW, s_.stats)
                    sts.append(s_.stats)
                gn = (sts, norm.weight.detach(), norm.bias.detach(), norm.num_groups, norm.eps, act)
            else:
                x = norm(x, act=act, x2=x2)
                x2 = None
        splits = None
        if x2 is not None:
            splits = [x.C, x2.C]
            assert x.C + x2.C == self.in_channels, (x.C, x2.C, self.in_channels)
        wp, b = self.packed(dtype, splits)
        OH, OW = self.out_hw(x.H, x.W, upsample)
        pad = 0 if self.asym_pad else self.padding
        st = ops.new_chsums(x.B, self.n_pad, x.t.device) if (stats and not ops.EXP_NO_STATS) else None
        post_gn = None
        if post_norm is not None:
            assert self.n_pad == self.out_channels and residual is None
            post_gn = (post_norm.weight.detach(), post_norm.bias.detach(), post_norm.num_groups, post_norm.eps, True)
        o = ops.conv2d(x.t, wp, x.B, x.H, x.W, N=self.n_pad, x2=None if x2 is None else x2.t,
                       KH=self.kernel_size, KW=self.kernel_size, stride=self.stride, pad_t=pad, pad_l=pad,
                       OH=OH, OW=OW, upsample=upsample, bias=b, rowvec=rowvec,
                       residual=None if residual is None else residual.t, out=out,
                       alg_nk=(self.out_channels, self.kernel_size * self.kernel_size * self.in_channels), stats=st,
                       gn=gn, post_gn=post_gn)
        if post_norm is not None:
            o, applied = o
            if applied:
                return Tok(o, x.B, OH, OW)
            return post_norm(Tok(o, x.B, OH, OW, st), silu=True)
        return Tok(o, x.B, OH, OW, st)


class Linear(_Packed):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None

    def packed(self, dtype):
        def build():
            w = self.weight.detach().float()
            b = None if self.bias is None else self.bias.detach().float().contiguous()
            return packing.pack_linear_weight(w, dtype, ops.k_tile(dtype)).to(w.device), b

        return self._cache_get((dtype, "lin"), build)

    def forward(self, x, residual=None, out=None, stats=None, B=1):
        """x: [M, K] dense 2-D tensor of the compute dtype."""
        wp, b = self.packed(x.dtype)
        return ops.linear(x, wp, bias=b, residual=residual, out=out, stats=stats, B=B)


class GroupNorm(nn.Module):
    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.empty(num_channels))
        self.bias = nn.Parameter(torch.empty(num_channels))

    def forward(self, x, silu=False, x2=None, act=None, residual=None):
        """act(GroupNorm([x | x2]) + residual); ``act``: None/'silu'/'relu' (``silu=True`` is the short form);
        returns one dense Tok."""
        xs = [x.t] if x2 is None else [x.t, x2.t]
        st = [x.stats] if x2 is None else [x.stats, x2.stats]
        t = ops.groupnorm(xs, x.B, x.HW, self.num_groups, self.weight.detach(), self.bias.detach(), self.eps,
                          silu=silu, stats=st, act=act, residual=None if residual is None else residual.t)
        return x.like(t)


class LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.empty(dim))
        self.bias = nn.Parameter(torch.empty(dim))

    def forward(self, x):
        return ops.layernorm(x, self.weight.detach(), self.bias.detach(), self.eps)


class Identity(nn.Module):
    """Parameter-free placeholder keeping diffusers' module indices (Dropout, SiLU)."""

    def forward(self, x):
        return x
