"""``DAFormerHead`` on the HIP kernels: /root/reference/modeling/sem_seg_head/daformer_head.py
:536-749 with the shipped decoder_params (``embed_cfg = mlp``, ``fusion_cfg = aspp, sep=True, dilations
(1, 6, 12, 18), pool=False, BN + ReLU``; config_files/common/models/mtmadise_multi_lora.py:42-64).

Parameter names follow mmcv 1.3.7 ``ConvModule`` / ``DepthwiseSeparableConvModule`` (``conv``, ``bn``,
``depthwise_conv``, ``pointwise_conv``) so reference checkpoints load.  Inference only: BatchNorm uses its
running statistics (folded into the conv weights / the depthwise epilogue) and Dropout2d is the identity --
the train-mode head (per-GPU batch statistics) belongs to the backward rows.
"""
import torch
import torch.nn as nn

from . import ops, packing
from ._lib import EPI_RELU, EPI_NONE, ACT_RELU
from .nn import Tok, Linear, _Packed
from .backbone import widen_tokens


class _BN(nn.Module):
    """BatchNorm2d parameter/buffer container (eval-mode affine: y = x * s + t)."""

    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def affine(self):
        s = self.weight.detach().float() / torch.sqrt(self.running_var.float() + self.eps)
        return s, self.bias.detach().float() - self.running_mean.float() * s


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

    def forward(self, x, out=None):
        dtype = x.t.dtype

        def build():
            s, t = self.bn.affine()
            w = self.conv.weight.detach().float() * s[:, None, None, None]
            return packing.pack_conv_weight(w, dtype, ops.k_tile(dtype)), t.contiguous()

        wp, b = self._cache_get((dtype,), build)
        o = ops.conv2d(x.t, wp, x.B, x.H, x.W, N=self.cout, KH=self.k, KW=self.k, pad_t=self.padding, pad_l=self.padding,
                       bias=b, epilogue=EPI_RELU, out=out)
        return Tok(o, x.B, x.H, x.W)


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

    def forward(self, x):
        def build():
            s, t = self.bn.affine()
            w9c = self.conv.weight.detach().float().reshape(self.c, 9).t().contiguous()   # [9][C]
            return w9c, s.contiguous(), t.contiguous()

        w9c, s, t = self._cache_get(("dw",), build)
        return Tok(ops.dwconv3x3(x.t, w9c, s, t, x.B, x.H, x.W, self.dilation, ACT_RELU), x.B, x.H, x.W)


class DepthwiseSeparableConvModule(nn.Module):
    def __init__(self, cin, cout, k, dilation, padding):
        super().__init__()
        assert k == 3 and padding == dilation
        self.depthwise_conv = _DepthwiseConvModule(cin, dilation)
        self.pointwise_conv = ConvModule(cin, cout, 1)

    def forward(self, x, out=None):
        return self.pointwise_conv(self.depthwise_conv(x), out=out)


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

    def forward(self, x):
        n = len(self.dilations)
        cat = torch.empty((x.t.shape[0], n * self.channels), dtype=x.t.dtype, device=x.t.device)
        for i, m in enumerate(self.aspp_modules):   # every branch writes its column window of the concat buffer
            m(x, out=cat[:, i * self.channels:(i + 1) * self.channels])
        return self.bottleneck(x.like(cat))


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
        self.conv_seg = nn.Conv2d(channels, num_classes, kernel_size=1)   # parameter container (diffusers-style name)
        self.dropout = nn.Identity()

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
        x = self._tokens(input_features)
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
        h = self.fuse_layer(x0.like(cat))
        # cls_seg: Dropout2d is the identity in eval; 1x1 conv to the classes (N padded to a multiple of 4)
        K = self.num_classes
        Kp = (K + 3) // 4 * 4
        w, b = self._cls_weights(h.t.dtype, Kp)
        logits = ops.conv2d(h.t, w, h.B, h.H, h.W, N=Kp, bias=b)
        return ops.nhwc_to_nchw(logits, h.B, K, h.H, h.W)

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
            self.__dict__["_cls_cache"] = c
        return c[1], c[2]
