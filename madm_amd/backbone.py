"""Backbone-side rows of the hot path on the HIP kernels:

* ``ClipFeatureProject`` / ``BasePromptTimeGenerator`` -- /root/reference/modeling/meta_arch/ldm_base.py:632-717,720-968
  (learnable prompt / time conditioning; the shipped ``clip_state='no'`` path);
* ``BottleneckBlock`` -- detectron2's ``BottleneckBlock(norm="GN")`` as instantiated by
  modeling/backbone/feature_extractor.py:347-359 (``ResNet.make_stage``);
* ``AttentionFeatureExtractorBackbone`` -- modeling/backbone/feature_extractor.py:20-284 (base class) and :287-396.

Same constructor arguments, attribute names, parameter names and return structures as the reference; the
arithmetic runs through libmadm_hip.  detectron2's ``Backbone`` base class is not required: this class
provides the members the reference's meta-arch and checkpointer touch.
"""
import math
from collections import OrderedDict, defaultdict

import torch
import torch.nn as nn

from . import ops
from .nn import Tok, Conv2d, GroupNorm


class FeatureDict(dict):
    """{'s0': NCHW f32 tensor, ...} as the reference returns it, plus ``.tok``: the same features as
    channels-last ``Tok``s for the HIP segmentation head (no layout round trip)."""
    tok = None


def widen_tokens(t, C, width):
    """[M, >=C] tokens -> zero-padded [M, width] buffer holding the first C channels (tiny C only)."""
    if t.shape[1] == width:
        return t
    buf = torch.zeros((t.shape[0], width), dtype=t.dtype, device=t.device)
    ops.copy_columns(t, buf, C)
    return buf


# ----------------------------------------------------------------------------- a2: prompt / time conditioning
class ClipFeatureProject(nn.Module):
    """ldm_base.py:632-717 for ``input_prefix=False`` (clip_state == 'no', the shipped setting)."""

    def __init__(self, learnable_cond_prompt=False, prompt_in_features=None, prompt_out_features=None,
                 prompt_seq_len=None, learnable_cond_time=False, time_in_features=None, time_out_features=None,
                 time_seq_len=None, time_alpha_cond_size=None, input_prefix=False, without_prompt_alpha=True,
                 multi_layer_prompt=False, init_uncond_prompt=False, uncond_prompt=None):
        super().__init__()
        if input_prefix or multi_layer_prompt or init_uncond_prompt:
            raise NotImplementedError("ClipFeatureProject: input_prefix / multi_layer_prompt / init_uncond_prompt are "
                                      "not enabled by any shipped config (SURVEY.md Appendix C.11)")
        self.learnable_cond_prompt = learnable_cond_prompt
        self.learnable_cond_time = learnable_cond_time
        self.input_prefix = input_prefix
        self.without_prompt_alpha = without_prompt_alpha
        self.multi_layer_prompt = multi_layer_prompt
        self.init_uncond_prompt = init_uncond_prompt
        if learnable_cond_prompt:
            pe = torch.zeros(1, prompt_seq_len, prompt_out_features)
            nn.init.trunc_normal_(pe, std=0.02)
            self.prompt_embed = nn.Parameter(pe)
            if not without_prompt_alpha:
                shape = [1, prompt_seq_len, prompt_out_features]
                self.alpha_cond_prompt = nn.Parameter(torch.rand(shape))
                self.alpha_uncond_prompt = nn.Parameter(torch.rand(shape))
        if learnable_cond_time:
            self.alpha_cond_time = nn.Parameter(torch.zeros(time_alpha_cond_size))
            te = torch.zeros(1, time_seq_len, time_out_features)
            nn.init.trunc_normal_(te, std=0.02)
            self.time_embed = nn.Parameter(te)

    def get_cond_prompt(self, uncond_prompt, prefix=None, repeat=1):
        if not self.learnable_cond_prompt:
            return ops.tanh_gate(uncond_prompt.contiguous(), repeat=repeat)
        if self.without_prompt_alpha:
            return ops.tanh_gate(self.prompt_embed.detach(), repeat=repeat)
        assert uncond_prompt.shape[1] == self.alpha_cond_prompt.shape[1], \
            "prompt length != uncond prompt length needs the bilinear prompt resize (ldm_base.py:677-678): not built"
        # tanh(alpha_u) * uncond + tanh(alpha_c) * prompt_embed  (:681)
        return ops.tanh_gate(uncond_prompt.contiguous(), self.alpha_uncond_prompt.detach(), self.prompt_embed.detach(),
                             self.alpha_cond_prompt.detach(), repeat=repeat)

    def get_cond_time(self, prefix=None, repeat=1):
        if not self.learnable_cond_time:
            return None
        # tanh(alpha_t) * time_embed  (:706); alpha is [1280], time_embed [1, 1, 1280]
        return ops.tanh_gate(self.time_embed.detach(), self.alpha_cond_time.detach().view_as(self.time_embed), repeat=repeat)

    def forward(self, uncond_prompt, prefix=None, repeat=1):
        return self.get_cond_prompt(uncond_prompt, prefix, repeat), self.get_cond_time(prefix, repeat)

    def backward(self, uncond_prompt, dcond_inputs, dcond_emb, grads=None):
        """Gradients of :meth:`forward`'s two outputs (f32 [B, 77, 768] / [B, 1280] or None) w.r.t. the learnable prompt /
        time parameters, ACCUMULATED into ``grads`` {parameter name: f32 tensor} (created when None) -- the two gradient
        passes of a training step share these parameters when ``same_cond_params`` (ldm_base.py:632-717)."""
        grads = {} if grads is None else grads

        def acc(name, like):
            if name not in grads:
                grads[name] = torch.zeros_like(like, dtype=torch.float32)
            return grads[name]

        if dcond_inputs is not None and self.learnable_cond_prompt:
            d = dcond_inputs.float().contiguous()
            if self.without_prompt_alpha:
                ops.tanh_gate_backward(d, self.prompt_embed.detach(), dx1=acc("prompt_embed", self.prompt_embed))
            else:
                ops.tanh_gate_backward(d, uncond_prompt.contiguous(), self.alpha_uncond_prompt.detach(),
                                       self.prompt_embed.detach(), self.alpha_cond_prompt.detach(),
                                       da1=acc("alpha_uncond_prompt", self.alpha_uncond_prompt),
                                       da2=acc("alpha_cond_prompt", self.alpha_cond_prompt),
                                       dx2=acc("prompt_embed", self.prompt_embed))
        if dcond_emb is not None and self.learnable_cond_time:
            d = dcond_emb.float().contiguous()
            ops.tanh_gate_backward(d, self.time_embed.detach().contiguous(), self.alpha_cond_time.detach().contiguous(),
                                   da1=acc("alpha_cond_time", self.alpha_cond_time), dx1=acc("time_embed", self.time_embed))
        return grads


class BasePromptTimeGenerator(nn.Module):
    """ldm_base.py:720-968 with ``ldm_extractor`` given (the LazyConfig always passes one)."""

    cross_attention_out_dim = [320, 320, 640, 640, 1280, 1280, 1280, 1280, 1280, 1280, 640, 640, 640, 320, 320, 320]

    def __init__(self, learnable_cond_prompt=True, learnable_cond_time=True, same_cond_params=False,
                 detach_prompt_for_mixed_data=False, clip_state="no", num_timesteps=1, clip_model_name="",
                 ldm_extractor=None, without_prompt_alpha=False, multi_layer_prompt=False,
                 mix_source_target_prompt=False, init_uncond_prompt=False, mask_prompt_ratio=False,
                 detach_mask_prompt=False, prompt_perturbation=False, rand_prompt_scale=None, **kwargs):
        super().__init__()
        if ldm_extractor is None:
            raise NotImplementedError("the legacy CompVis LdmExtractor fallback (ldm_base.py:763-768) is out of scope")
        if clip_state != "no":
            raise NotImplementedError("clip_state != 'no' (image-conditioned prompts, ldm_base.py:776-780) is not built")
        for flag, name in ((multi_layer_prompt, "multi_layer_prompt"), (mix_source_target_prompt, "mix_source_target_prompt"),
                           (init_uncond_prompt, "init_uncond_prompt"), (mask_prompt_ratio, "mask_prompt_ratio"),
                           (prompt_perturbation, "prompt_perturbation"), (rand_prompt_scale, "rand_prompt_scale")):
            if flag:
                raise NotImplementedError(f"BasePromptTimeGenerator option {name} (SURVEY.md Appendix C.11) is not built")
        self.learnable_cond_prompt = learnable_cond_prompt
        self.learnable_cond_time = learnable_cond_time
        self.same_cond_params = same_cond_params
        self.detach_prompt_for_mixed_data = detach_prompt_for_mixed_data
        self.clip_state = clip_state
        self.multi_layer_prompt = multi_layer_prompt
        self.without_prompt_alpha = without_prompt_alpha
        self.ldm_extractor = ldm_extractor
        self.text_embed_shape = ldm_extractor.text_embed_shape
        tdim = ldm_extractor.unet_time_embed_out_features
        self.uncond_inputs = ldm_extractor.uncond_inputs.detach()
        prompt_seq_len = kwargs.get("prompt_seq_len", self.text_embed_shape[0])

        def make():
            return ClipFeatureProject(
                learnable_cond_prompt=learnable_cond_prompt, prompt_in_features=None,
                prompt_out_features=self.text_embed_shape[1], prompt_seq_len=prompt_seq_len,
                without_prompt_alpha=without_prompt_alpha, learnable_cond_time=learnable_cond_time,
                time_in_features=None, time_out_features=tdim, time_seq_len=num_timesteps, time_alpha_cond_size=tdim,
                input_prefix=False).to(self.uncond_inputs.device)

        self.clip_project_rgb = make()
        self.clip_project_others = self.clip_project_rgb if same_cond_params else make()

    def forward(self, batched_inputs, input_modal, ema_forward=False, timestep=None, return_unet_feats=False, **kwargs):
        assert input_modal in {'rgb', 'others', 'mixed', 'masked_prompt', 'prompt_perturbation', 'rand_prompt'}
        image = batched_inputs["img"]
        B = image.shape[0]
        if input_modal == 'rgb':
            assert ema_forward is False
            project = self.clip_project_rgb
        else:
            project = self.ema_clip_project_others if ema_forward else self.clip_project_others
        # the reference computes [1, 77, 768] / [1, 1, 1280] and repeat_interleaves to the batch (:915-917);
        # the gate kernel writes the B copies directly
        cond_inputs, cond_emb = project(self.uncond_inputs, None, repeat=B)
        batched_inputs["cond_inputs"] = cond_inputs
        batched_inputs["cond_emb"] = cond_emb
        assert not return_unet_feats
        if timestep is not None:
            batched_inputs['timestep'] = timestep
        return self.ldm_extractor(batched_inputs, input_modal, ema_forward=ema_forward, **kwargs)

    feature_size = property(lambda self: self.ldm_extractor.feature_size)
    feature_dims = property(lambda self: self.ldm_extractor.feature_dims)
    feature_strides = property(lambda self: self.ldm_extractor.feature_strides)
    num_groups = property(lambda self: self.ldm_extractor.num_groups)
    grouped_indices = property(lambda self: self.ldm_extractor.grouped_indices)

    def set_requires_grad(self, requires_grad):
        for p in self.ldm_extractor.unet.parameters():
            p.requires_grad = requires_grad


# ----------------------------------------------------------------------------- a8: projections
class _D2Conv(Conv2d):
    """detectron2 ``Conv2d`` wrapper: conv (no bias) with its norm as the ``.norm`` sub-module."""

    def __init__(self, cin, cout, k, padding=0):
        super().__init__(cin, cout, k, padding=padding, bias=False)
        self.norm = GroupNorm(32, cout, eps=1e-5)


class BottleneckBlock(nn.Module):
    """detectron2 BottleneckBlock(norm="GN", stride 1): relu(GN(1x1)) -> relu(GN(3x3)) -> GN(1x1), plus
    GN(1x1 shortcut) when the widths differ, relu(sum)."""

    def __init__(self, in_channels, out_channels, *, bottleneck_channels, norm="GN"):
        super().__init__()
        assert norm == "GN"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.shortcut = _D2Conv(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.conv1 = _D2Conv(in_channels, bottleneck_channels, 1)
        self.conv2 = _D2Conv(bottleneck_channels, bottleneck_channels, 3, padding=1)
        self.conv3 = _D2Conv(bottleneck_channels, out_channels, 1)

    def forward(self, x, tape=None):
        """``tape`` (a list): receives the record :meth:`backward` needs -- the raw conv outputs with their channel sums;
        the normalised / activated tensors are recomputed there."""
        kt = ops.k_tile(x.t.dtype)
        if x.C != self.in_channels:   # e.g. the 3-channel decoder image travelling in a 4-wide tensor
            assert x.C > self.in_channels
        xin = x
        if x.C % kt != 0:
            xin = Tok(widen_tokens(x.t, self.in_channels, (self.in_channels + kt - 1) // kt * kt), x.B, x.H, x.W)
        h1 = self.conv1(xin)
        h2 = self.conv2(h1, norm=self.conv1.norm, act="relu")        # GN + ReLU folded into the 3x3 conv when it fits
        a2 = self.conv2.norm(h2, act="relu")
        h3 = self.conv3(a2)
        hs = None
        if self.shortcut is not None:
            hs = self.shortcut(xin)
            s = self.shortcut.norm(hs)
        else:
            s = x
        out = self.conv3.norm(h3, act="relu", residual=s)
        if tape is not None:
            tape.append((self, xin, h1, h2, h3, hs, out))
        return out

    def backward(self, rec, dout, need_dx=True):
        """Backward of the recorded forward for the output gradient ``dout`` [M, Cout]: returns (dx [M, Cin(padded)] or
        None, {parameter name: f32 gradient}).  torch autograd through detectron2's BottleneckBlock in the reference
        (feature_extractor.py:347-359,367-396)."""
        from . import backward as bw
        _, xin, h1, h2, h3, hs, out = rec
        B, HW = xin.B, xin.HW
        grads = {}

        def gn_bwd(conv, h, dy, act, dres=None):
            n = conv.norm
            (dh,), dg, db = ops.groupnorm_backward([h.t], dy, B, HW, n.num_groups, n.weight.detach().float(),
                                                   n.bias.detach().float(), n.eps, [h.stats], act=act, dres=dres)
            return dh, dg, db

        def put(name, g):
            for k_, v in g.items():
                grads[name + "." + k_] = v

        dsum = ops.relu_backward(out.t, dout.contiguous())           # d(GN(conv3) + shortcut)
        dh3, grads["conv3.norm.weight"], grads["conv3.norm.bias"] = gn_bwd(self.conv3, h3, dsum, "none")
        a2 = self.conv2.norm(h2, act="relu")
        da2, g = bw.conv2d_backward(self.conv3, a2, dh3)
        put("conv3", g)
        dh2, grads["conv2.norm.weight"], grads["conv2.norm.bias"] = gn_bwd(self.conv2, h2, da2, "relu")
        a1 = self.conv1.norm(h1, act="relu")
        da1, g = bw.conv2d_backward(self.conv2, a1, dh2)
        put("conv2", g)
        dh1, grads["conv1.norm.weight"], grads["conv1.norm.bias"] = gn_bwd(self.conv1, h1, da1, "relu")
        dx, g = bw.conv2d_backward(self.conv1, xin, dh1, need_dx=need_dx)
        put("conv1", g)
        if self.shortcut is not None:
            dhs, grads["shortcut.norm.weight"], grads["shortcut.norm.bias"] = gn_bwd(self.shortcut, hs, dsum, "none")
            dx, g = bw.conv2d_backward(self.shortcut, xin, dhs, need_dx=need_dx, dres=dx)
            put("shortcut", g)
        elif need_dx:
            dx = ops.add(dx, dsum)
        return dx, grads


class FeatureExtractorBackbone(nn.Module):
    """feature_extractor.py:20-284 (the members the shipped configs use)."""

    def __init__(self, feature_extractor, out_features, backbone_in_size=(512, 512), min_stride=4, max_stride=32,
                 projection_dim=512, num_res_blocks=1, use_checkpoint=False, slide_training=False,
                 slide_inference=False):
        super().__init__()
        self.feature_extractor = feature_extractor
        self.use_checkpoint = use_checkpoint
        if slide_training:
            raise NotImplementedError("slide_training needs the training path (not built)")
        # sliding-window inference: three 512-wide windows over a 512 x 1024 input (feature_extractor.py:73-75)
        self.y1_y2_x1_x2 = [(0, 512, 0, 512), (0, 512, 256, 768), (0, 512, 512, 1024)] if slide_inference else None
        if isinstance(projection_dim, int):
            self.feature_projections = nn.ModuleList()
            for feature_dim in self.feature_extractor.feature_dims:
                self.feature_projections.append(nn.Sequential(*[
                    BottleneckBlock(feature_dim if i == 0 else projection_dim, projection_dim,
                                    bottleneck_channels=projection_dim // 4) for i in range(num_res_blocks)]))
        self._slide_inference = slide_inference
        self._slide_training = slide_training
        self.backbone_in_size = tuple(backbone_in_size)
        self.min_stride, self.max_stride = min_stride, max_stride
        idx_to_stride, stride_to_indices = {}, defaultdict(list)
        for indices in self.feature_extractor.grouped_indices:
            for idx in indices:
                stride = min(max(self.feature_extractor.feature_strides[idx], min_stride), max_stride)
                idx_to_stride[idx] = stride
                stride_to_indices[stride].append(idx)
        self._sorted_grouped_indices = [stride_to_indices[s] for s in sorted(stride_to_indices)]
        self._out_feature_channels, self._out_feature_strides = {}, {}
        for indices in self._sorted_grouped_indices:
            stride = idx_to_stride[indices[0]]
            name = f"s{int(math.log2(stride))}"
            if name not in out_features:
                continue
            assert name not in self._out_feature_strides, f"Duplicate feature name {name}"
            self._out_feature_strides[name] = stride
            self._out_feature_channels[name] = projection_dim
        self._out_features = list(self._out_feature_strides.keys())

    @property
    def size_divisibility(self):
        return 64

    def ignored_state_dict(self, destination=None, prefix=""):
        if destination is None:
            destination = OrderedDict()
            destination._metadata = OrderedDict()
        for name, module in self._modules.items():
            if module is not None and hasattr(module, "ignored_state_dict"):
                module.ignored_state_dict(destination, prefix + name + ".")
        return destination

    def preprocess_image(self, img):
        """T.Resize(backbone_in_size, bilinear) then zero-pad to a multiple of 64 (feature_extractor.py:77-79,140-146).
        Identity for 512x512 inputs."""
        H, W = self.backbone_in_size
        if tuple(img.shape[-2:]) != (H, W):
            img = ops.resize_bilinear_nchw(img.float().contiguous(), H, W)
        assert H % self.size_divisibility == 0 and W % self.size_divisibility == 0, \
            "backbone_in_size is a multiple of 64 in every shipped config"
        return img

    def checkpoint_forward_features(self, features, input_image_size, ema_forward=False):
        return self.forward_features(features, input_image_size, ema_forward)

    def single_forward(self, img, input_modal='rgb', ema_forward=False, timestep=None, **kwargs):
        input_image_size = img.shape[-2:]
        img = self.preprocess_image(img)
        features = self.feature_extractor(dict(img=img), input_modal, ema_forward, timestep, _return_tokens=True, **kwargs)
        if 'return_unet_final_output' in kwargs.keys():
            return self.checkpoint_forward_features(features[0], input_image_size, ema_forward), features[1]
        return self.checkpoint_forward_features(features, input_image_size, ema_forward)

    def slide_forward(self, img, input_modal='rgb', ema_forward=False, timestep=None, **kwargs):
        """feature_extractor.py:199-278: the windows are cropped on the device, run as ONE batched forward
        (images are independent units: GroupNorm / LayerNorm are per-sample) and their projected features are
        averaged into the full-size canvas by madm_slide_merge (sum of the covering windows / their count)."""
        if 'return_unet_final_output' in kwargs:
            raise NotImplementedError("return_unet_final_output with sliding windows is a training-time combination")
        B, _, h_img, w_img = img.shape
        short = min(h_img, w_img)
        wins = self.y1_y2_x1_x2
        h_grids = max(h_img - short + short - 1, 0) // short + 1
        w_grids = max(w_img - short + short - 1, 0) // short + 1 + 1
        assert h_grids == 1 and w_grids == 3 == len(wins), "the reference supports exactly three windows (:233)"
        img = img.float().contiguous()
        crops = torch.empty((len(wins) * B, img.shape[1], short, short), dtype=torch.float32, device=img.device)
        for k, (y1, y2, x1, x2) in enumerate(wins):
            assert (y2 - y1, x2 - x1) == (short, short)
            ops.scale_pad_nchw(img, 1.0, short, short, y1, x1, out=crops[k * B:(k + 1) * B])
        feats = self.single_forward(crops, input_modal, ema_forward, timestep, **kwargs)['output_features']
        out_tok = {}
        for name, f in feats.tok.items():
            stride = self._out_feature_strides[name]
            assert f.B == len(wins) * B
            merged = ops.slide_merge(f.t, len(wins), B, f.H, f.W, w_img // stride, [w[2] // stride for w in wins])
            out_tok[name] = Tok(merged, B, h_img // stride, w_img // stride)
        res = FeatureDict({k: v.nchw() for k, v in out_tok.items()})
        res.tok = out_tok
        return {'output_features': res}

    def forward(self, img, input_modal='rgb', ema_forward=False, timestep=None, **kwargs):
        if not self._slide_inference:
            return self.single_forward(img, input_modal, ema_forward, timestep, **kwargs)
        return self.slide_forward(img, input_modal, ema_forward, timestep, **kwargs)


class AttentionFeatureExtractorBackbone(FeatureExtractorBackbone):
    """feature_extractor.py:287-396: one BottleneckBlock projection per selected scale; features are routed
    by their WIDTH (``features_dict[i.shape[-1]]``, ``res = 512 // stride``, :371-373,383)."""

    def __init__(self, attention_features_res, feature_dims, attention_features_location, target_attention_loss=False,
                 attention_select_index=None, feature_extractor=None, out_features=None, backbone_in_size=(512, 512),
                 min_stride=4, max_stride=32, projection_dim=(512, 512, 512, 512), bottleneck_channels=512 // 4,
                 num_res_blocks=1, use_checkpoint=False, slide_training=False, slide_inference=False):
        super().__init__(feature_extractor, out_features, backbone_in_size, min_stride, max_stride, list(projection_dim),
                         num_res_blocks, use_checkpoint, slide_training, slide_inference)
        self.attention_features_res = attention_features_res
        self.feature_dims = feature_dims
        self.attention_features_location = attention_features_location
        self.target_attention_loss = target_attention_loss
        self.attention_select_index = attention_select_index
        self.feature_projections = nn.ModuleList()
        for i, feature_dim in enumerate(self.feature_dims):
            self.feature_projections.append(nn.Sequential(*[
                BottleneckBlock(feature_dim if j == 0 else projection_dim[i], projection_dim[i],
                                bottleneck_channels=bottleneck_channels) for j in range(num_res_blocks)]))
        self._out_feature_strides = {s: 2 ** int(s[1]) for s in out_features}
        self._out_features = list(self._out_feature_strides.keys())

    def forward_features(self, features, input_image_size, ema_forward=False):
        self.attention_features = dict()
        features_dict = {}
        for f in features:   # Tok (HIP path) or NCHW tensor
            if not isinstance(f, Tok):
                B, C, H, W = f.shape
                f = Tok(ops.nchw_to_nhwc(f.float().contiguous(), self.feature_extractor.ldm_extractor.compute_dtype,
                                         C if C % 64 == 0 else (C + 63) // 64 * 64), B, H, W)
            features_dict[f.W] = f
        out_tok = {}
        if self.feature_extractor.ldm_extractor.final_fuse_vae_decoder_feat:
            out_tok['s0'] = features_dict[512]
        projections = self.ema_feature_projections if ema_forward else self.feature_projections
        tape = getattr(self, "_tape", None)      # set by forward_features_recorded (the training step)
        for idx, name in enumerate(self._out_features):
            res = self.feature_res(name)
            h = features_dict[res]
            recs = [] if tape is not None else None
            for blk in projections[idx]:
                h = blk(h, tape=recs)
            if tape is not None:
                tape.append((idx, name, res, recs))
            out_tok[name] = h
        if tape is not None:   # training: the head consumes the tokens, no NCHW copies
            feats = FeatureDict()
        else:
            feats = FeatureDict({k: v.nchw() for k, v in out_tok.items()})
        feats.tok = out_tok
        return {'output_features': feats}

    def feature_res(self, name):
        """Width of the extractor feature that feeds output ``name``: ``512 // stride`` in the reference
        (feature_extractor.py:383, 512 = backbone_in_size); generalised to the configured backbone_in_size."""
        return self.backbone_in_size[1] // self._out_feature_strides[name]

    def forward_features_recorded(self, features, input_image_size, tape):
        """:meth:`forward_features` with the projections' backward records appended to ``tape`` (student projections)."""
        self._tape = tape
        try:
            return self.forward_features(features, input_image_size, ema_forward=False)
        finally:
            self._tape = None

    def backward_features(self, tape, dfeats, no_grad_inputs=()):
        """dfeats: {output name: gradient tokens [M, C_out]} -> ({feature width: gradient tokens of that extractor
        feature}, {parameter name (relative to the backbone): f32 gradient}).  Inputs whose width is in
        ``no_grad_inputs`` (the detached VAE-decoder image behind 's0', ldm_diffusers.py:196-201) get no data gradient."""
        dins, grads = {}, {}
        for idx, name, res, recs in tape:
            d = dfeats[name]
            need_dx = res not in no_grad_inputs
            for j in reversed(range(len(recs))):
                d, g = recs[j][0].backward(recs[j], d, need_dx=need_dx or j > 0)
                for k_, v in g.items():
                    grads[f"feature_projections.{idx}.{j}.{k_}"] = v
            if need_dx:
                assert res not in dins
                dins[res] = d
        return dins, grads
