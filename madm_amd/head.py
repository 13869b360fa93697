"""``DAFormerHead`` on the HIP kernels: /root/reference/modeling/sem_seg_head/daformer_head.py
:536-749 with the shipped decoder_params (``embed_cfg = mlp``, ``fusion_cfg = aspp, sep=True, dilations
(1, 6, 12, 18), pool=False, BN + ReLU``; config_files/common/models/mtmadise_multi_lora.py:42-64).

Parameter names follow mmcv 1.3.7 ``ConvModule`` / ``DepthwiseSeparableConvModule`` (``conv``, ``bn``,
``depthwise_conv``, ``pointwise_conv``) so reference checkpoints load.

Two modes, selected by ``nn.Module.training`` exactly as torch's BatchNorm2d / Dropout2d do:
* eval: BatchNorm uses its running statistics (folded into the conv weights / the depthwise epilogue), Dropout2d is the
  identity;
* train (the self-training step AND its EMA teacher, which the reference never puts into eval mode, cmdise.py:307-335):
  per-GPU batch statistics (never SyncBN, mtmadise_multi_lora.py:49,61) -- over channels-last
  tokens a train-mode BatchNorm2d is the GroupNorm kernel with one image of B*H*W pixels and one group per channel --
  running statistics updated with momentum 0.1, ``Dropout2d(0.1)`` as a per-(image, channel) scale.  ``forward_tokens``
  records what :meth:`DAFormerHead.backward_tokens` needs (raw conv outputs + their batch sums; normalised tensors are
  recomputed), the backward returns the feature gradients and ``{parameter name: f32 gradient}``.
"""
import torch
import torch.nn as nn

from . import ops, packing
from ._lib import EPI_RELU, EPI_NONE, ACT_RELU, ACT_NONE
from .nn import Tok, Linear, _Packed
from .backbone import widen_tokens


class _BN(nn.Module):
    """BatchNorm2d parameter/buffer container (eval-mode affine: y = x * s + t; train mode: ops.batchnorm_train)."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def affine(self):
        s = self.weight.detach().float() / torch.sqrt(self.running_var.float() + self.eps)
        return s, self.bias.detach().float() - self.running_mean.float() * s

    def train_apply(self, h, chsums, act, out=None):
        """Train-mode forward on raw conv output tokens h [M, C]: batch statistics (from the conv epilogue's per-image
        sums when given), running-statistic update, normalise (+act).  Returns (y, batch sums st)."""
        st = ops.batch_stats(h, chsums, running=(self.running_mean, self.running_var), momentum=self.momentum)
        # the kernel wrote the running statistics through raw pointers: move their version counters like an in-place
        # torch op would, so the eval-mode folded operands are re-derived
        torch.autograd.graph.increment_version((self.running_mean, self.running_var))
        self.num_batches_tracked += 1
        y = ops.batchnorm_train(h, st, self.weight.detach(), self.bias.detach(), self.eps, act=act, out=out)
        return y, st

    def train_backward(self, h, st, dy, act):
        return ops.batchnorm_backward(h, st, dy, self.weight.detach(), self.bias.detach(), self.eps, act=act)


class _RawConv(nn.Module):
    def __init__(self, cin, cout, k, groups=1, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None


class ConvModule(_Packed):
    """mmcv ConvModule(conv(bias=False) -> BN -> ReLU); dense conv with BN folded into the packed weights."""

    def __init__(self, cin, cout, k, padding=0, dilation=1):
        super().__init__()
        assert dilation == 1
        self.cin, self.cout, self.k, self.padding = cin, cout, k, padding
        self.conv = _RawConv(cin, cout, k)
        self.bn = _BN(cout)
        self.activate = nn.Identity()

    def _versions(self):
        ts = [self.conv.weight, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        return tuple((t._version, t.data_ptr()) for t in ts)

    def _wver(self):
        return (self.conv.weight._version, self.conv.weight.data_ptr())

    def _raw(self, dtype):
        return self._cache_get((dtype, "raw"), lambda: packing.pack_conv_weight(self.conv.weight.detach().float(), dtype,
                                                                                 ops.k_tile(dtype)), ver=self._wver())

    def forward(self, x, out=None, tape=None):
        dtype = x.t.dtype
        if self.training:
            sums = ops.new_chsums(x.B, self.cout, x.t.device)
            h = ops.conv2d(x.t, self._raw(dtype), x.B, x.H, x.W, N=self.cout, KH=self.k, KW=self.k, pad_t=self.padding,
                           pad_l=self.padding, stats=sums)
            y, st = self.bn.train_apply(h, sums, "relu", out=out)
            if tape is not None:
                tape.append((self, x, h, st))
            return Tok(y, x.B, x.H, x.W)

        def build():
            s, t = self.bn.affine()
            w = self.conv.weight.detach().float() * s[:, None, None, None]
            return packing.pack_conv_weight(w, dtype, ops.k_tile(dtype)), t.contiguous()

        wp, b = self._cache_get((dtype,), build)
        o = ops.conv2d(x.t, wp, x.B, x.H, x.W, N=self.cout, KH=self.k, KW=self.k, pad_t=self.padding, pad_l=self.padding,
                       bias=b, epilogue=EPI_RELU, out=out)
        return Tok(o, x.B, x.H, x.W)

    def backward(self, rec, dy, grads, prefix, dres=None):
        """rec = the tape entry of the train-mode forward; dy = gradient of the (windowed) output.  Returns dx [M, Cin]
        (+ dres); parameter gradients go into ``grads``."""
        _, x, h, st = rec
        dtype = x.t.dtype
        dh, dg, db = self.bn.train_backward(h, st, dy, "relu")
        grads[prefix + "bn.weight"], grads[prefix + "bn.bias"] = dg, db
        k, p = self.k, self.padding
        dwp = ops.conv2d_wgrad(x.t, dh, x.B, x.H, x.W, KH=k, KW=k, pad_t=p, pad_l=p)
        grads[prefix + "conv.weight"] = packing.unpack_conv_weight_grad(dwp, self.cin, k, k, ops.k_tile(dtype))
        wt = self._cache_get((dtype, "dgrad"), lambda: ops.pack_dgrad_weights(self._raw(dtype), k * k), ver=self._wver())
        return ops.conv2d_dgrad(dh, wt, x.B, x.H, x.W, C=wt.shape[0], KH=k, KW=k, pad_t=p, pad_l=p, residual=dres)


class _DepthwiseConvModule(_Packed):
    def __init__(self, c, dilation):
        super().__init__()
        self.c, self.dilation = c, dilation
        self.conv = _RawConv(c, c, 3, groups=c)
        self.bn = _BN(c)
        self.activate = nn.Identity()

    def _versions(self):
        ts = [self.conv.weight, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        return tuple((t._version, t.data_ptr()) for t in ts)

    def _raw(self):
        def build():
            w9c = self.conv.weight.detach().float().reshape(self.c, 9).t().contiguous()   # [9][C]
            one = torch.ones(self.c, device=w9c.device)
            return w9c, w9c.flip(0).contiguous(), one, torch.zeros_like(one)
        return self._cache_get(("dw_raw",), build, ver=(self.conv.weight._version, self.conv.weight.data_ptr()))

    def forward(self, x, tape=None):
        if self.training:
            w9c, _, one, zero = self._raw()
            h = ops.dwconv3x3(x.t, w9c, one, zero, x.B, x.H, x.W, self.dilation, ACT_NONE)
            y, st = self.bn.train_apply(h, None, "relu")
            if tape is not None:
                tape.append((self, x, h, st))
            return Tok(y, x.B, x.H, x.W)

        def build():
            s, t = self.bn.affine()
            w9c = self.conv.weight.detach().float().reshape(self.c, 9).t().contiguous()   # [9][C]
            return w9c, s.contiguous(), t.contiguous()

        w9c, s, t = self._cache_get(("dw",), build)
        return Tok(ops.dwconv3x3(x.t, w9c, s, t, x.B, x.H, x.W, self.dilation, ACT_RELU), x.B, x.H, x.W)

    def backward(self, rec, dy, grads, prefix):
        _, x, h, st = rec
        dh, dg, db = self.bn.train_backward(h, st, dy, "relu")
        grads[prefix + "bn.weight"], grads[prefix + "bn.bias"] = dg, db
        dw9c = ops.dwconv3x3_wgrad(x.t, dh, x.B, x.H, x.W, self.dilation)
        grads[prefix + "conv.weight"] = dw9c.t().reshape(self.c, 1, 3, 3).contiguous()
        _, wflip, one, zero = self._raw()   # data gradient: the same dilated depthwise conv with the taps reversed
        return ops.dwconv3x3(dh, wflip, one, zero, x.B, x.H, x.W, self.dilation, ACT_NONE)


class DepthwiseSeparableConvModule(nn.Module):
    def __init__(self, cin, cout, k, dilation, padding):
        super().__init__()
        assert k == 3 and padding == dilation
        self.depthwise_conv = _DepthwiseConvModule(cin, dilation)
        self.pointwise_conv = ConvModule(cin, cout, 1)

    def forward(self, x, out=None, tape=None):
        return self.pointwise_conv(self.depthwise_conv(x, tape=tape), out=out, tape=tape)


class MLP(nn.Module):
    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        self.proj = Linear(input_dim, embed_dim)


class ASPPWrapper(nn.Module):
    def __init__(self, in_channels, channels, sep, dilations, pool, norm_cfg=None, act_cfg=None, align_corners=False,
                 context_cfg=None):
        super().__init__()
        if pool or context_cfg is not None or not sep:
            raise NotImplementedError("only the shipped sep-ASPP (pool=False, no context layer) is built")
        self.dilations = tuple(dilations)
        self.channels = channels
        mods = []
        for d in self.dilations:
            mods.append(ConvModule(in_channels, channels, 1) if d == 1
                        else DepthwiseSeparableConvModule(in_channels, channels, 3, dilation=d, padding=d))
        self.aspp_modules = nn.ModuleList(mods)
        self.bottleneck = ConvModule(len(self.dilations) * channels, channels, 3, padding=1)

    def forward(self, x, tape=None):
        n = len(self.dilations)
        cat = torch.empty((x.t.shape[0], n * self.channels), dtype=x.t.dtype, device=x.t.device)
        for i, m in enumerate(self.aspp_modules):   # every branch writes its column window of the concat buffer
            m(x, out=cat[:, i * self.channels:(i + 1) * self.channels], tape=tape)
        return self.bottleneck(x.like(cat), tape=tape)

    def backward(self, tape, dy, grads, prefix):
        """Reverse of the train-mode forward over its tape (entries in forward order); returns dx [M, Cin]."""
        recs = list(tape)
        dcat = self.bottleneck.backward(recs.pop(), dy, grads, prefix + "bottleneck.")
        dx = None
        for i in reversed(range(len(self.aspp_modules))):
            m = self.aspp_modules[i]
            win = dcat[:, i * self.channels:(i + 1) * self.channels]
            if isinstance(m, DepthwiseSeparableConvModule):
                rec_pw = recs.pop()
                rec_dw = recs.pop()
                dmid = m.pointwise_conv.backward(rec_pw, win, grads, f"{prefix}aspp_modules.{i}.pointwise_conv.")
                d = m.depthwise_conv.backward(rec_dw, dmid, grads, f"{prefix}aspp_modules.{i}.depthwise_conv.")
                dx = d if dx is None else ops.add(dx, d)
            else:   # the 1x1 branch adds the others' sum through its data-gradient GEMM's residual input
                dx = m.backward(recs.pop(), win, grads, f"{prefix}aspp_modules.{i}.", dres=dx)
        assert not recs
        return dx


class DAFormerHead(nn.Module):
    def __init__(self, in_channels, in_keys, channels, *, num_classes, dropout_ratio=0.1, conv_cfg=None, norm_cfg=None,
                 act_cfg=None, in_index=-1, input_transform='multiple_select', decoder_params=None, ignore_index=255,
                 align_corners=False, init_cfg=None, concat_attention_to_conv_seg=False,
                 final_fuse_vae_decoder_feat=False):
        super().__init__()
        if concat_attention_to_conv_seg or final_fuse_vae_decoder_feat:
            raise NotImplementedError("concat_attention_to_conv_seg / final_fuse_vae_decoder_feat (no shipped config) are "
                                      "not built")
        assert input_transform == 'multiple_select' and not align_corners
        self.in_channels, self.in_index, self.in_keys = list(in_channels), list(in_index), list(in_keys)
        self.channels, self.num_classes, self.dropout_ratio = channels, num_classes, dropout_ratio
        self.ignore_index, self.align_corners = ignore_index, align_corners
        self.final_fuse_vae_decoder_feat = False
        embed_dims = decoder_params['embed_dims']
        if isinstance(embed_dims, int):
            embed_dims = [embed_dims] * len(self.in_index)
        embed_cfg = dict(decoder_params['embed_cfg'])
        neck = decoder_params['embed_neck_cfg']
        neck = embed_cfg if neck == 'same_as_embed_cfg' else dict(neck)
        fusion_cfg = dict(decoder_params['fusion_cfg'])
        if embed_cfg.get('type') != 'mlp' or neck.get('type') != 'mlp' or fusion_cfg.pop('type') != 'aspp':
            raise NotImplementedError("only embed 'mlp' + fusion 'aspp' (the shipped decoder_params) are built")
        self.embed_dims = embed_dims
        self.embed_layers = nn.ModuleDict({str(i): MLP(c, e) for i, c, e in zip(self.in_index, self.in_channels, embed_dims)})
        fusion_cfg.pop('align_corners', None)
        self.fuse_layer = ASPPWrapper(in_channels=sum(embed_dims), channels=channels, **fusion_cfg)
        self.conv_seg = nn.Conv2d(channels, num_classes, kernel_size=1)   # parameter container (mmseg's name)
        self.dropout = nn.Identity()                                      # Dropout2d(dropout_ratio): see _dropout_scale
        self.dropout_generator = None       # torch.Generator for the Dropout2d masks (None: the device's default RNG)
        self.dropout_scale_override = None  # tests: a fixed f32 [B, channels] scale (keep / (1 - p)) instead of a draw

    def transfer_input_dict_to_list(self, inputs_dict):
        lst = [inputs_dict[k] for k in self.in_keys]
        assert len(lst) == len(inputs_dict)   # "ensure all features from backbone are used"
        return lst

    def _tokens(self, input_features):
        tok = getattr(input_features, "tok", None)
        if tok is not None:
            return [tok[k] for k in self.in_keys]
        out = []
        for k in self.in_keys:   # NCHW tensors from a non-HIP backbone
            f = input_features[k]
            B, C, H, W = f.shape
            dtype = torch.bfloat16 if self._dtype is None else self._dtype
            out.append(Tok(ops.nchw_to_nhwc(f.float().contiguous(), dtype, C), B, H, W))
        return out

    _dtype = None

    def forward(self, input_dict):
        """{'output_features': {key: feature}} -> logits [B, num_classes, H0, W0] f32 at the first feature's
        size (daformer_head.py:702-749); features are Toks (HIP backbone) or NCHW tensors."""
        input_features = input_dict['output_features']
        self.transfer_input_dict_to_list(input_features)
        logits = self.forward_tokens(self._tokens(input_features))
        return ops.nhwc_to_nchw(logits.t, logits.B, self.num_classes, logits.H, logits.W)

    def _dropout_scale(self, B, device):
        """nn.Dropout2d(p): whole (image, channel) planes are zeroed with probability p, the rest scaled by 1 / (1 - p)."""
        if self.dropout_scale_override is not None:
            s = self.dropout_scale_override
            if isinstance(s, list):          # a FIFO: one scale per call
                s = s.pop(0)
            s = s.to(device=device, dtype=torch.float32).contiguous()
            assert tuple(s.shape) == (B, self.channels)
            return s
        p = float(self.dropout_ratio)
        keep = torch.rand((B, self.channels), device=device, generator=self.dropout_generator) >= p
        return keep.to(torch.float32) / (1.0 - p)

    def forward_tokens(self, x, tape=None):
        """list of feature Toks (in ``in_keys`` order) -> f32 logit tokens Tok [B*H0*W0, Kp] (first num_classes columns
        valid).  Train mode: batch-statistic BatchNorm + Dropout2d; ``tape`` (a dict) receives what
        :meth:`backward_tokens` needs."""
        x0 = x[0]
        M0 = x0.t.shape[0]
        cat = torch.empty((M0, sum(self.embed_dims)), dtype=x0.t.dtype, device=x0.t.device)
        off = 0
        for i, e in zip(self.in_index, self.embed_dims):
            f = x[i]
            win = cat[:, off:off + e]
            lin = self.embed_layers[str(i)].proj
            if (f.H, f.W) == (x0.H, x0.W):
                lin(f.t, out=win)
            else:   # MLP at the feature's own resolution, then bilinear to the first feature's size (:737-746)
                ops.resize_bilinear(lin(f.t), f.B, f.H, f.W, x0.H, x0.W, out=win)
            off += e
        fuse_tape = [] if tape is not None else None
        h = self.fuse_layer(x0.like(cat), tape=fuse_tape) if self.training else self.fuse_layer(x0.like(cat))
        # cls_seg: Dropout2d (identity in eval), then the 1x1 conv to the classes (N padded to a multiple of 4), f32 out
        K = self.num_classes
        Kp = (K + 3) // 4 * 4
        scale = None
        hd = h.t
        if self.training and self.dropout_ratio > 0:
            scale = self._dropout_scale(h.B, h.t.device)
            hd = ops.scale_channels(h.t, scale, h.B, h.HW)
        w, b = self._cls_weights(h.t.dtype, Kp)
        logits = ops.conv2d(hd, w, h.B, h.H, h.W, N=Kp, bias=b, out_f32=True)
        if tape is not None:
            assert self.training, "the head's backward is the train-mode one"
            tape.update(x=list(x), fuse=fuse_tape, hd=hd, scale=scale, geom=(h.B, h.H, h.W))
        return Tok(logits, h.B, h.H, h.W)

    def backward_tokens(self, tape, dlogits):
        """dlogits: [M0, ld >= num_classes] tokens of the compute dtype (columns beyond num_classes zero).  Returns
        ([d feature_i tokens, dense [M_i, C_i(padded)]] in ``in_keys`` order, {parameter name: f32 gradient})."""
        from . import backward as bw
        x, hd, scale = tape["x"], tape["hd"], tape["scale"]
        B, H, W = tape["geom"]
        dtype = hd.dtype
        K, C = self.num_classes, self.channels
        grads = {}
        # ---- conv_seg ----
        N = dlogits.shape[1]
        db = ops.zeros_f32((N,), dlogits.device)
        dwp = ops.conv2d_wgrad(hd, dlogits, B, H, W, dbias=db)                        # [N, C]; bias sums by the same launch
        grads["conv_seg.weight"] = dwp[:K, :C].reshape(K, C, 1, 1).contiguous()
        grads["conv_seg.bias"] = db[:K].contiguous()
        wt = self._cls_dgrad_weights(dtype, N)
        dhd = ops.conv2d_dgrad(dlogits, wt, B, H, W, C=wt.shape[0])
        dh = dhd if scale is None else ops.scale_channels(dhd, scale, B, H * W)
        # ---- fusion (sep-ASPP + bottleneck) ----
        dcat = self.fuse_layer.backward(tape["fuse"], dh, grads, "fuse_layer.")
        # ---- embedding MLPs (+ bilinear resize) ----
        x0 = x[0]
        dfeats, off = [], 0
        for i, e in zip(self.in_index, self.embed_dims):
            f = x[i]
            win = dcat[:, off:off + e]
            lin = self.embed_layers[str(i)].proj
            if (f.H, f.W) != (x0.H, x0.W):
                win = ops.resize_bilinear_backward(win, f.B, f.H, f.W, x0.H, x0.W)
            df, g = bw.linear_backward(lin, f.t, win)
            for k_, v in g.items():
                grads[f"embed_layers.{i}.proj.{k_}"] = v
            dfeats.append(df)
            off += e
        return dfeats, grads

    def _cls_dgrad_weights(self, dtype, N):
        key = (dtype, N, self.conv_seg.weight._version, self.conv_seg.weight.data_ptr())
        c = self.__dict__.get("_cls_dgrad_cache")
        if c is None or c[0] != key:
            with torch.no_grad():
                w = self.conv_seg.weight.detach().float()
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, N - w.shape[0]))
                wp = packing.pack_conv_weight(w, dtype, ops.k_tile(dtype))              # [N, C]
                c = (key, ops.pack_dgrad_weights(wp, 1))
            ops.note_build()
            self.__dict__["_cls_dgrad_cache"] = c
        return c[1]

    def _cls_weights(self, dtype, Kp):
        key = (dtype, self.conv_seg.weight._version, self.conv_seg.weight.data_ptr())
        c = self.__dict__.get("_cls_cache")
        if c is None or c[0] != key:
            with torch.no_grad():
                w = self.conv_seg.weight.detach().float()
                b = self.conv_seg.bias.detach().float()
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, Kp - w.shape[0]))
                b = torch.nn.functional.pad(b, (0, Kp - b.shape[0])).contiguous()
                c = (key, packing.pack_conv_weight(w, dtype, ops.k_tile(dtype)), b)
            ops.note_build()
            self.__dict__["_cls_cache"] = c
        return c[1], c[2]
