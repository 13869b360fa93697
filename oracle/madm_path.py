"""ORACLE (test infrastructure only): the full inference forward of the shipped MADM configuration on CPU fp32.

``build_reference_eval_model`` instantiates, IN THE BUILD CONTAINER, the reference's own classes loaded by path
(oracle/ref_driver.load_modeling): ``BasePromptTimeGenerator`` (modeling/meta_arch/ldm_base.py:720-968),
``AttentionFeatureExtractorBackbone`` (modeling/backbone/feature_extractor.py:287-396) and ``DAFormerHead``
(modeling/sem_seg_head/daformer_head.py:536-749) with the arguments of
config_files/common/models/mtmadise_multi_lora.py:13-64 + the per-task overrides of
config_files/SemSeg/MTMADISE/mtmadise_cityscapes_rgb_to_depth_11.py:10-55, on top of ``OracleLdm``.
``build_oracle_eval_model`` builds the same graph from the oracle's own restatements (usable anywhere).
``eval_forward`` restates MTMADISE.forward's eval branch (modeling/meta_arch/mtmadise.py:657-691)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import sd_modules, ldm_path, third_party as tp

DEPTH_CFG = dict(out_features=["s0", "s3", "s4", "s5"], feature_dims=[3, 320, 640, 1280],
                 projection_dim=[128, 512, 512, 512], head_in_channels=[128, 512, 512, 512], num_classes=11,
                 vae_decoder_loss=True)
# config_files/SemSeg/MTMADISE/mtmadise_cityscapes_rgb_to_infrared_9.py: the same graph with the 9 FMB classes
INFRARED_CFG = dict(DEPTH_CFG, num_classes=9)
S345_CFG = dict(out_features=["s3", "s4", "s5"], feature_dims=[320, 640, 1280], projection_dim=[512, 512, 512],
                head_in_channels=[512, 512, 512], num_classes=11, vae_decoder_loss=False)   # main.py:480-484


def cfg_by_name(name):
    return {"DEPTH": DEPTH_CFG, "INFRARED": INFRARED_CFG, "S345": S345_CFG}[name]


def head_decoder_params():
    return dict(embed_dims=256, embed_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                embed_neck_cfg=dict(type='mlp', act_cfg=None, norm_cfg=None),
                fusion_cfg=dict(type='aspp', sep=True, dilations=(1, 6, 12, 18), pool=False, act_cfg=dict(type='ReLU'),
                                norm_cfg=dict(type='BN', requires_grad=True)))


def uncond_stand_in():
    """The seeded stand-in LdmRocm uses for CLIP('') when no snapshot is available (madm_amd/ldm_rocm.py)."""
    return 0.02 * torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(4242))


def make_oracle_ldm(vae, unet, cfg):
    return ldm_path.OracleLdm(vae, unet, sd_modules.DDPMScheduler(), uncond_stand_in(), encoder_block_indices=[],
                              unet_block_indices=[5, 8, 11], decoder_block_indices=(), input_range='-1+1',
                              unet_block_indices_type='after', vae_decoder_loss=cfg["vae_decoder_loss"])


def build_reference_eval_model(ns, vae, unet, cfg):
    """ns = ref_driver.load_modeling().  Returns (backbone, head) built from the REFERENCE classes."""
    ldm = make_oracle_ldm(vae, unet, cfg)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self       # ldm_base.py:774 calls .cuda(); CPU-only container
    try:
        gen = ns.ldm_base.BasePromptTimeGenerator(learnable_cond_prompt=True, learnable_cond_time=True, clip_state='no',
                                                  num_timesteps=1, clip_model_name="ViT-L-14-336", ldm_extractor=ldm,
                                                  same_cond_params=True)
    finally:
        torch.Tensor.cuda = orig_cuda
    backbone = ns.feature_extractor.AttentionFeatureExtractorBackbone(
        attention_features_res=None, feature_dims=list(cfg["feature_dims"]), projection_dim=list(cfg["projection_dim"]),
        attention_features_location=None, feature_extractor=gen, num_res_blocks=1,
        out_features=list(cfg["out_features"]), use_checkpoint=False, slide_training=False)
    n = len(cfg["out_features"])
    head = ns.daformer_head.DAFormerHead(
        in_channels=list(cfg["head_in_channels"]), in_keys=list(cfg["out_features"]), in_index=list(range(n)), channels=256,
        dropout_ratio=0.1, num_classes=cfg["num_classes"], norm_cfg=dict(type='BN', requires_grad=True),
        align_corners=False, decoder_params=head_decoder_params())
    return backbone.eval(), head.eval()


# ---- the oracle's own restatement of the same classes (no /root/reference needed) ------------------------
class ClipFeatureProject(nn.Module):
    """ldm_base.py:632-717, input_prefix=False."""

    def __init__(self, seq_len=77, dim=768, tdim=1280, without_prompt_alpha=False):
        super().__init__()
        self.without_prompt_alpha = without_prompt_alpha
        self.prompt_embed = nn.Parameter(tp.trunc_normal_(torch.zeros(1, seq_len, dim), std=0.02))
        if not without_prompt_alpha:
            self.alpha_cond_prompt = nn.Parameter(torch.rand(1, seq_len, dim))
            self.alpha_uncond_prompt = nn.Parameter(torch.rand(1, seq_len, dim))
        self.alpha_cond_time = nn.Parameter(torch.zeros(tdim))
        self.time_embed = nn.Parameter(tp.trunc_normal_(torch.zeros(1, 1, tdim), std=0.02))

    def forward(self, uncond_prompt, prefix=None):
        if self.without_prompt_alpha:
            prompt = self.prompt_embed
        else:
            prompt = torch.tanh(self.alpha_uncond_prompt) * uncond_prompt + torch.tanh(self.alpha_cond_prompt) * self.prompt_embed
        return prompt, torch.tanh(self.alpha_cond_time) * self.time_embed


class PromptTimeGenerator(nn.Module):
    """ldm_base.py:832-924 for clip_state='no', same_cond_params=True."""

    def __init__(self, ldm):
        super().__init__()
        self.ldm_extractor = ldm
        self.uncond_inputs = ldm.uncond_inputs.detach()
        self.clip_project_rgb = ClipFeatureProject()
        self.clip_project_others = self.clip_project_rgb

    def forward(self, batched_inputs, input_modal, ema_forward=False, timestep=None, **kwargs):
        project = self.clip_project_rgb if input_modal == 'rgb' else \
            (self.ema_clip_project_others if ema_forward else self.clip_project_others)
        ci, ce = project(self.uncond_inputs, None)
        B = batched_inputs["img"].shape[0]
        batched_inputs["cond_inputs"], batched_inputs["cond_emb"] = ci, ce
        if timestep is not None:
            batched_inputs['timestep'] = timestep
        if B != 1:
            batched_inputs["cond_inputs"] = torch.repeat_interleave(ci, B, dim=0)
            batched_inputs["cond_emb"] = torch.repeat_interleave(ce, B, dim=0)
        return self.ldm_extractor(batched_inputs, input_modal, ema_forward=ema_forward, **kwargs)


class OracleBackbone(nn.Module):
    """feature_extractor.py:156-170,367-396."""

    def __init__(self, feature_extractor, cfg):
        super().__init__()
        self.feature_extractor = feature_extractor
        self.feature_projections = nn.ModuleList([
            nn.Sequential(*tp.ResNet.make_stage(tp.BottleneckBlock, num_blocks=1, in_channels=fd, bottleneck_channels=128,
                                                out_channels=pd, norm="GN"))
            for fd, pd in zip(cfg["feature_dims"], cfg["projection_dim"])])
        self._out_features = list(cfg["out_features"])
        self._strides = {s: 2 ** int(s[1]) for s in self._out_features}
        self.resize = tp.Resize((512, 512))

    def forward(self, img, input_modal='rgb', ema_forward=False, timestep=None, **kwargs):
        img = tp.ImageList.from_tensors(list(self.resize(img)), 64).tensor
        features = self.feature_extractor(dict(img=img), input_modal, ema_forward, timestep, **kwargs)
        fd = {f.shape[-1]: f for f in features}
        proj = self.ema_feature_projections if ema_forward else self.feature_projections
        return {'output_features': {n: proj[i](fd[512 // self._strides[n]]) for i, n in enumerate(self._out_features)}}


class OracleHead(nn.Module):
    """daformer_head.py:536-749 with the shipped decoder_params."""

    def __init__(self, cfg):
        super().__init__()
        n = len(cfg["out_features"])
        self.in_keys = list(cfg["out_features"])
        self.embed_layers = nn.ModuleDict({str(i): _MLP(c, 256) for i, c in enumerate(cfg["head_in_channels"])})
        norm, act = dict(type='BN'), dict(type='ReLU')
        self.fuse_layer = _ASPP(256 * n, 256, norm, act)
        self.conv_seg = nn.Conv2d(256, cfg["num_classes"], 1)
        self.dropout = nn.Dropout2d(0.1)

    def forward(self, input_dict):
        x = [input_dict['output_features'][k] for k in self.in_keys]
        n = x[-1].shape[0]
        os_size = x[0].shape[2:]
        cs = []
        for i, f in enumerate(x):
            c = self.embed_layers[str(i)](f).permute(0, 2, 1).contiguous().reshape(n, -1, f.shape[2], f.shape[3])
            if c.shape[2:] != os_size:
                c = F.interpolate(c, size=os_size, mode='bilinear', align_corners=False)
            cs.append(c)
        return self.conv_seg(self.dropout(self.fuse_layer(torch.cat(cs, dim=1))))


class _MLP(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.proj = nn.Linear(cin, cout)

    def forward(self, x):
        return self.proj(x.flatten(2).transpose(1, 2).contiguous())


class _ASPP(nn.Module):
    def __init__(self, cin, ch, norm, act, dilations=(1, 6, 12, 18)):
        super().__init__()
        self.aspp_modules = nn.ModuleList([
            tp.ConvModule(cin, ch, 1, norm_cfg=norm, act_cfg=act) if d == 1 else
            tp.DepthwiseSeparableConvModule(cin, ch, 3, dilation=d, padding=d, norm_cfg=norm, act_cfg=act)
            for d in dilations])
        self.bottleneck = tp.ConvModule(len(dilations) * ch, ch, kernel_size=3, padding=1, norm_cfg=norm, act_cfg=act)

    def forward(self, x):
        return self.bottleneck(torch.cat([m(x) for m in self.aspp_modules], dim=1))


def build_oracle_eval_model(vae, unet, cfg):
    ldm = make_oracle_ldm(vae, unet, cfg)
    return OracleBackbone(PromptTimeGenerator(ldm), cfg).eval(), OracleHead(cfg).eval()


@torch.no_grad()
def eval_forward(backbone, head, image_0_255, eval_with_noise=None):
    """mtmadise.py:657-691 for one image [3, H, W] in 0..255: returns (sem_seg [1, K, H, W], backbone features)."""
    x = (image_0_255 - 0.0) / 255.0
    ori = x.shape[1:]
    t = tp.ImageList.from_tensors([x], 64).tensor
    kw = {'input_modal': 'others'}
    if eval_with_noise is not None:
        kw['timestep'] = (eval_with_noise, eval_with_noise + 1)
    feats = backbone(t, **kw)
    out = head(feats)
    out = F.interpolate(out, size=t.shape[2:], mode='bilinear', align_corners=False)
    return out[:, :, :ori[0], :ori[1]], feats


# ---- evaluator (evaluation/d2_evaluator.py:99-127, 240-275), numpy restatement -------------------------------------
def evaluator_confusion(pred, gt, num_classes, ignore_label=255):
    """pred / gt: int arrays; returns the (K+1) x (K+1) int64 confusion matrix exactly as the reference builds it."""
    import numpy as np
    pred = np.array(pred, dtype=np.int32)
    gt = np.int32(np.array(gt)).copy()
    gt[gt == ignore_label] = num_classes
    n = num_classes + 1
    return np.bincount(n * pred.reshape(-1) + gt.reshape(-1), minlength=n * n).reshape(n, n).astype(np.int64)


def evaluator_metrics(conf, num_classes):
    import numpy as np
    acc = np.full(num_classes, np.nan, dtype=np.float64)
    iou = np.full(num_classes, np.nan, dtype=np.float64)
    tp = conf.diagonal()[:-1].astype(np.float64)
    pos_gt = np.sum(conf[:-1, :-1], axis=0).astype(np.float64)
    class_weights = pos_gt / np.sum(pos_gt)
    pos_pred = np.sum(conf[:-1, :-1], axis=1).astype(np.float64)
    acc_valid = pos_gt > 0
    acc[acc_valid] = tp[acc_valid] / pos_gt[acc_valid]
    iou_valid = (pos_gt + pos_pred) > 0
    union = pos_gt + pos_pred - tp
    iou[acc_valid] = tp[acc_valid] / union[acc_valid]
    return {"mIoU": 100 * np.sum(iou[acc_valid]) / np.sum(iou_valid),
            "fwIoU": 100 * np.sum(iou[acc_valid] * class_weights[acc_valid]),
            "mACC": 100 * np.sum(acc[acc_valid]) / np.sum(acc_valid), "pACC": 100 * np.sum(tp) / np.sum(pos_gt),
            "iou": iou, "acc": acc}


def slide_forward(backbone, img):
    """feature_extractor.py:199-278 on an OracleBackbone: three 512-wide windows over a 512 x 1024 input, features summed
    into the canvas and divided by the per-pixel window count."""
    wins = [(0, 512, 0, 512), (0, 512, 256, 768), (0, 512, 512, 1024)]
    B, _, h_img, w_img = img.shape
    out, cnt = {}, {}
    for (y1, y2, x1, x2) in wins:
        feats = backbone(img[:, :, y1:y2, x1:x2], input_modal='others')['output_features']
        for k, f in feats.items():
            s = backbone._strides[k]
            if k not in out:
                out[k] = torch.zeros((B, f.shape[1], h_img // s, w_img // s))
                cnt[k] = torch.zeros_like(out[k])
            out[k][:, :, y1 // s:y2 // s, x1 // s:x2 // s] += f
            cnt[k][..., y1 // s:y2 // s, x1 // s:x2 // s] += 1
    assert all((c == 0).sum() == 0 for c in cnt.values())
    return {'output_features': {k: out[k] / cnt[k] for k in out}}
