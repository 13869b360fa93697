"""ORACLE (test infrastructure only -- never imported by madm_amd): CPU fp32 restatement of ONE TRAINING STEP of the
shipped RGB->Depth configuration, differentiable through torch autograd.

``OracleMTMADISE.forward_train`` follows /root/reference/modeling/meta_arch/mtmadise.py:180-656 line by line for the flags
of config_files/SemSeg/MTMADISE/mtmadise_cityscapes_rgb_to_depth_11.py (lora_configs = [], vae_decoder_loss = 'st' / 'L1',
reg_uncertain, rev_noise_sup + gradually, enable_mixup) on top of the oracle's backbone / head restatements
(oracle/madm_path.py, pinned against the reference's own classes by the eval fixtures) and

* ``CmdiseCriterion``: the REFERENCE's own class when /root/reference is present (``reference_criterion()``, loaded by
  path: modeling/criterion.py imports only torch / numpy), else ``CmdiseCriterionRestated`` below, which
  tests/test_oracle.py pins against the reference class on seeded logits (SURVEY.md 8c pin K6);
* ``CMDISE._inti_ema_weights`` / ``_update_ema`` (modeling/meta_arch/cmdise.py:307-349);
* ``strong_transform`` restricted to ClassMix (utils/dacs_transforms.py:11-26,100-111; ``color_aug_flag=False`` -- the
  kornia colour jitter / blur is pinned separately, oracle/augment.py).

LoRA branch (``lora_configs`` non-empty, the north_star variant; no shipped config enables it, main.py:613-616,794 does):
``'name_rN_aM'`` parsing (:48-54), ``set_multi_lora`` (:115-127: one adapter per name on to_k / to_q / to_v / to_out.0, all
set active, ``ldm_extractor._freeze()`` re-applied), ``set_lora_adapter`` before every pass (:240 'default' for the source
pass; :286, :310 the target modality for the mixed-image and teacher passes; :672 eval), ``add_zero_gead_on_unused_lora``
(:149-157, :654-655).  ``ldm_path.OracleLdm._freeze`` restates ldm_diffusers.py:101-121: every extractor parameter frozen, then the
``finetune_unet`` selection -- so with ``finetune_unet='no'`` the freshly added adapters are frozen as well (that IS the
reference's behaviour: peft's ``requires_grad=True`` on the adapters is overwritten by the ``_freeze()`` call of :127).

Deviations, all stated: (1) the backbone's hard-coded 512 (``T.Resize((512, 512))``, ``res = 512 // stride``,
feature_extractor.py:77-79,383) is the parameter ``in_size`` so that the step can run at 64 x 64 in seconds -- the 512
routing itself is pinned by tests/golden/eval_depth.npz from the reference's class; (2) ``nn.Dropout2d`` draws from torch's
CPU generator, which no device kernel can reproduce: the (image, channel) keep-scales are INJECTED (``FixedDropout2d``),
one per head call in call order; (3) ``reg_uncertain``'s distance map only feeds the visualisation (:323-328 ->
:556-560) and is not computed.
"""
import importlib.util
import os
import random
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import labels as OL
from . import ldm_path, madm_path, third_party as tp

REF_CRITERION = "/root/reference/modeling/criterion.py"


def reference_criterion():
    """The reference's own CmdiseCriterion class (this container only)."""
    spec = importlib.util.spec_from_file_location("_madm_ref_criterion", REF_CRITERION)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.CmdiseCriterion


class CmdiseCriterionRestated(nn.Module):
    """modeling/criterion.py:110-254 for reduction='mean', class_weight=None: source / target cross entropy (:166-187) and
    the vae_decoder_loss entries (:236-246)."""

    def __init__(self, num_classes=19, pseudo_threshold=0.968, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert reduction == 'mean' and class_weight is None
        self.loss_weight = loss_weight

    @staticmethod
    def cross_entropy(pred, label, pixel_weight=None, ignore_index=255):
        loss = F.cross_entropy(pred, label, weight=None, reduction='none', ignore_index=ignore_index)   # :123
        if pixel_weight is not None:
            loss = loss * pixel_weight.float()                                                          # :126-129, :97
        return loss.mean()                                                                              # :101 (mean over ALL pixels)

    def forward(self, outputs, targets, **kwargs):
        losses = {}
        sp = F.interpolate(outputs['source_rgb_pred'], size=targets['source_gt'].shape[-2:], mode='bilinear',
                           align_corners=False)
        losses['source_loss'] = self.loss_weight * self.cross_entropy(sp, targets['source_gt'][:, 0])
        tp_ = F.interpolate(outputs['target_sec_modal_pred'], size=targets['target_pl'].shape[-2:], mode='bilinear',
                            align_corners=False)
        losses['target_loss'] = self.loss_weight * self.cross_entropy(tp_, targets['target_pl'][:, 0],
                                                                      pixel_weight=targets['target_pw'])
        if 'vae_decoder_loss' in targets:
            for key, e in targets['vae_decoder_loss'].items():
                pred, gt, mask = e['pred'], e['gt'], e['mask']
                d = F.l1_loss(pred, gt, reduction='none') if e['loss_type'] == 'L1' else F.mse_loss(pred, gt, reduction='none')
                mask = F.interpolate(mask, size=gt.shape[-2:], mode='nearest').repeat(1, gt.shape[1], 1, 1)
                losses[f'vae_decoder_{key}_loss'] = torch.sum(d * mask) / d.numel() * e['loss_weight']
        return losses


class FixedDropout2d(nn.Module):
    """nn.Dropout2d with injected keep-scales: ``scales`` is a FIFO of f32 [B, C] tensors (keep / (1 - p))."""

    def __init__(self):
        super().__init__()
        self.scales = []

    def forward(self, x):
        if not self.training:
            return x
        s = self.scales.pop(0)
        return x * s[:, :, None, None]


class GradLdm(ldm_path.OracleLdm):
    """OracleLdm whose forward keeps the autograd graph of the UNet stage (the reference's LdmDiffusers.forward is not
    under no_grad; only vae_encoder / vae_decoder are, ldm_diffusers.py:282,313)."""

    def forward(self, batched_inputs, input_modal, **kwargs):
        return ldm_path.OracleLdm.forward.__wrapped__(self, batched_inputs, input_modal, **kwargs)


class TrainBackbone(madm_path.OracleBackbone):
    """madm_path.OracleBackbone with the reference's ``single_forward`` return contract for
    ``return_unet_final_output`` (feature_extractor.py:156-170) and the 512 generalised to ``in_size``."""

    def __init__(self, feature_extractor, cfg, in_size=512):
        super().__init__(feature_extractor, cfg)
        self.in_size = in_size
        self.resize = tp.Resize((in_size, in_size))

    def forward(self, img, input_modal='rgb', ema_forward=False, timestep=None, **kwargs):
        img = tp.ImageList.from_tensors(list(self.resize(img)), 64).tensor
        features = self.feature_extractor(dict(img=img), input_modal, ema_forward, timestep, **kwargs)
        extra = None
        if 'return_unet_final_output' in kwargs:
            features, extra = features
        fd = {f.shape[-1]: f for f in features}
        proj = self.ema_feature_projections if ema_forward else self.feature_projections
        out = {'output_features': {n: proj[i](fd[self.in_size // self._strides[n]]) for i, n in enumerate(self._out_features)}}
        return out if extra is None else (out, extra)


class OracleMTMADISE(nn.Module):
    def __init__(self, backbone, sem_seg_head, criterion, *, target_modality='Depth', train_palette, ema_alpha=0.999,
                 pseudo_threshold=0.968, vae_decoder_loss='st', vae_decoder_loss_type='L1', vae_decoder_loss_weight=(1.0, 1.0),
                 rev_noise_sup=True, rev_noise_end_iter=5000, rev_noise_gradually=True, denoise_timestep_range=(60, 61),
                 reg_uncertain=True, blur=True, color_jitter_strength=0.2, color_jitter_probability=0.2,
                 lora_configs=(), add_zero_grad=False, eval_with_noise=None, init_ema=True):
        super().__init__()
        self.backbone, self.sem_seg_head, self.criterion = backbone, sem_seg_head, criterion
        self.sem_seg_head_sec_modal = self.sem_seg_head                      # cmdise.py:153-156
        self.lora_configs = dict()                                           # :48-54
        for lora_config in lora_configs:
            name, rank, alpha = lora_config.split('_')
            assert name in {'default', 'Infrared', 'Depth', 'Event'}
            self.lora_configs[name] = dict(rank=int(rank[1:]), alpha=int(alpha[1:]))
        self.target_modality = target_modality
        self.add_zero_grad = add_zero_grad
        if len(self.lora_configs.keys()) != 0:                               # :58-59
            self.set_multi_lora()
        self.train_iter_index = 0
        self.ema_alpha, self.pseudo_threshold = ema_alpha, pseudo_threshold
        self.vae_decoder_loss, self.vae_decoder_loss_type = vae_decoder_loss, vae_decoder_loss_type
        self.vae_decoder_loss_weight = list(vae_decoder_loss_weight)
        self.rev_noise_sup, self.rev_noise_end_iter, self.rev_noise_gradually = rev_noise_sup, rev_noise_end_iter, rev_noise_gradually
        self.denoise_timestep_range = list(denoise_timestep_range)
        self.reg_uncertain, self.blur = reg_uncertain, blur
        self.color_jitter_strength, self.color_jitter_probability = color_jitter_strength, color_jitter_probability
        pal = list(train_palette)
        self.train_palette = pal + [0] * (768 - len(pal))                    # mtmadise.py:97-99
        self.reg_target_palette = list(self.train_palette)
        self.eval_with_noise = eval_with_noise
        if init_ema:
            self._inti_ema_weights()

    def set_multi_lora(self):                                               # :115-127
        from .sd_modules import LoraConfig
        ldm = self.backbone.feature_extractor.ldm_extractor
        for lora_name in self.lora_configs.keys():
            lora_config = LoraConfig(r=self.lora_configs[lora_name]['rank'], lora_alpha=self.lora_configs[lora_name]['alpha'],
                                     init_lora_weights="gaussian", target_modules=["to_k", "to_q", "to_v", "to_out.0"])
            ldm.unet.add_adapter(adapter_config=lora_config, adapter_name=lora_name)
        ldm.unet.set_adapter(list(self.lora_configs.keys()))
        ldm._freeze()

    def set_lora_adapter(self, state):                                      # :129-147
        if len(self.lora_configs.keys()) == 0:
            return
        if isinstance(state, str):
            state = [state]
        unet = self.backbone.feature_extractor.ldm_extractor.unet
        for _, module in unet.named_modules():
            if hasattr(module, '_active_adapter'):                           # isinstance(module, BaseTunerLayer)
                module._active_adapter = state

    def add_zero_gead_on_unused_lora(self, used_modal):                     # :149-157
        loss = []
        unet = self.backbone.feature_extractor.ldm_extractor.unet
        for name, p in unet.named_parameters():
            if 'lora' in name and used_modal not in name:
                loss.append(torch.sum(p))
        loss = sum(loss) * 0.
        return loss

    def _inti_ema_weights(self):                                            # cmdise.py:307-335
        self.backbone.ema_feature_projections = deepcopy(self.backbone.feature_projections)
        self.ema_sem_seg_head = deepcopy(self.sem_seg_head)
        self.ema_parms = [self.backbone.ema_feature_projections, self.ema_sem_seg_head]
        self.updated_parms = [self.backbone.feature_projections, self.sem_seg_head]
        fe = self.backbone.feature_extractor
        fe.ema_clip_project_others = deepcopy(fe.clip_project_others)
        self.ema_parms.append(fe.ema_clip_project_others)
        self.updated_parms.append(fe.clip_project_others)
        for m in self.ema_parms:
            for p in m.parameters():
                p.detach_()

    def _update_ema(self, it):                                              # cmdise.py:337-349
        alpha_teacher = min(1 - 1 / (it + 1), self.ema_alpha)
        for em, um in zip(self.ema_parms, self.updated_parms):
            for ema_param, param in zip(em.parameters(), um.parameters()):
                ema_param.data[:] = alpha_teacher * ema_param.data + (1 - alpha_teacher) * param.data

    def forward_train(self, batched_inputs):
        if self.train_iter_index > 0:                                        # :184-185
            self._update_ema(self.train_iter_index)
        source = tp.ImageList.from_tensors([(x['source_rgb'] - 0.0) / 255.0 for x in batched_inputs], 64).tensor
        target = tp.ImageList.from_tensors([(x['target_second_modality'] - 0.0) / 255.0 for x in batched_inputs], 64).tensor
        gt = tp.ImageList.from_tensors([x['source_label'] for x in batched_inputs], 64).tensor      # [B, 1, H, W]
        B = source.shape[0]
        strong_parameters = {'mix': None, 'color_jitter': random.uniform(0, 1), 'color_jitter_s': self.color_jitter_strength,
                             'color_jitter_p': self.color_jitter_probability,
                             'blur': random.uniform(0, 1) if self.blur else 0, 'mean': None, 'std': None}    # :217-225
        tmod = self.target_modality
        vae = self.backbone.feature_extractor.ldm_extractor.vae

        # source pred (:239-256)
        self.set_lora_adapter(state='default')                               # :240
        feats, source_out = self.backbone(source, return_unet_final_output=True, input_modal='rgb')
        source_pred = self.sem_seg_head(feats)
        source_color_gt, source_color_gt_mask = OL.convert_label_to_rgb(gt, self.reg_target_palette)
        source_color_gt_latent = ldm_path.vae_encoder(vae, source_color_gt, [])[0]

        # mixed image (:261-279), color_aug_flag=False
        with torch.no_grad():
            mix_masks = OL.get_class_masks(gt)
            mixed_img = torch.cat([OL.one_mix(mix_masks[i], data=torch.stack((source[i], target[i])))[0] for i in range(B)])

        # target pred (:284-302)
        self.set_lora_adapter(state=tmod)                                    # :286
        feats, target_out = self.backbone(mixed_img, return_unet_final_output=True, input_modal='mixed')
        target_pred = self.sem_seg_head(feats)

        # teacher (:308-392)
        with torch.no_grad():
            self.set_lora_adapter(state=tmod)                                # :310
            kw = dict(input_modal='others', ema_forward=True)
            if self.rev_noise_sup and self.train_iter_index <= self.rev_noise_end_iter:
                t_ = random.randint(self.denoise_timestep_range[0], self.denoise_timestep_range[1])
                if self.rev_noise_gradually:
                    t_ = int(t_ * (1 - self.train_iter_index / self.rev_noise_end_iter))
                kw['timestep'] = (t_, t_ + 1)
            low_res_feats, _ = self.backbone(target, return_unet_final_output=True, **kw)
            ema_logits = self.ema_sem_seg_head(low_res_feats)
            pseudo_prob, pseudo_label, pseudo_weight = OL.pseudo_labels(ema_logits, target.shape[2:], self.pseudo_threshold)
            gt_pixel_weight = torch.ones(pseudo_weight.shape)
            mixed_lbl = [None] * B
            mixed_seg_weight = pseudo_weight.clone()
            for i in range(B):
                _, mixed_lbl[i] = OL.one_mix(mix_masks[i], target=torch.stack((gt[i][0], pseudo_label[i])))
                _, w = OL.one_mix(mix_masks[i], target=torch.stack((gt_pixel_weight[i], pseudo_weight[i])))
                mixed_seg_weight[i] = w
            mixed_lbl = torch.cat(mixed_lbl)
        loss_input = {'source_rgb_pred': source_pred, 'target_sec_modal_pred': target_pred}
        loss_target = {'source_gt': gt, 'target_pl': mixed_lbl, 'target_pw': mixed_seg_weight, 'vae_decoder_loss': {}}
        if 's' in self.vae_decoder_loss:                                     # :609-616
            loss_target['vae_decoder_loss']['source'] = {
                'pred': source_out['before_vae.decoder'], 'gt': source_color_gt_latent, 'mask': source_color_gt_mask,
                'loss_weight': self.vae_decoder_loss_weight[0], 'loss_type': self.vae_decoder_loss_type}
        if 't' in self.vae_decoder_loss:                                     # :394-397, :617-624
            target_color_gt, target_color_gt_mask = OL.convert_label_to_rgb(mixed_lbl, self.reg_target_palette)
            target_color_gt_latent = ldm_path.vae_encoder(vae, target_color_gt, [])[0]
            target_color_gt_mask = target_color_gt_mask * pseudo_weight[:, None]
            loss_target['vae_decoder_loss']['target'] = {
                'pred': target_out['before_vae.decoder'], 'gt': target_color_gt_latent, 'mask': target_color_gt_mask,
                'loss_weight': self.vae_decoder_loss_weight[1], 'loss_type': self.vae_decoder_loss_type}
        losses = self.criterion(loss_input, loss_target)
        if self.add_zero_grad:                                               # :654-655
            losses['zero_grad'] = self.add_zero_gead_on_unused_lora(tmod)
        self.train_iter_index += 1
        self.last_step = dict(mixed_img=mixed_img, mixed_lbl=mixed_lbl, mixed_seg_weight=mixed_seg_weight,
                              pseudo_label=pseudo_label, pseudo_weight=pseudo_weight, ema_logits=ema_logits,
                              source_logits=source_pred, target_logits=target_pred)
        return losses


def _forward_eval(self, batched_inputs):
    """mtmadise.py:657-691: the eval branch, with the adapter switch of :672."""
    assert len(batched_inputs) == 1
    assert 'modality_type' not in batched_inputs[0].keys()
    target_modal_type = self.target_modality
    target_sec_modal = [(x['target_second_modality'] - 0.0) / 255.0 for x in batched_inputs]
    ori_size = target_sec_modal[0].shape[1:]
    target_sec_modal = tp.ImageList.from_tensors(target_sec_modal, 64)
    self.set_lora_adapter(state=target_modal_type)
    test_input_dict = {'input_modal': 'others'}
    if self.eval_with_noise is not None:
        test_input_dict['timestep'] = (self.eval_with_noise, self.eval_with_noise + 1)
    backbone_feats = self.backbone(target_sec_modal.tensor, **test_input_dict)
    outputs = self.sem_seg_head_sec_modal(backbone_feats)
    outputs = F.interpolate(outputs, size=target_sec_modal.tensor.shape[2:], mode='bilinear', align_corners=False)
    outputs = outputs[:, :, :ori_size[0], :ori_size[1]]
    self.last_eval_feats = backbone_feats
    return [{'sem_seg': outputs}]


OracleMTMADISE.forward_eval = torch.no_grad()(_forward_eval)


def build(vae, unet, cfg, criterion_cls=None, in_size=512, train_palette=None, finetune_unet='all', **kw):
    """OracleMTMADISE of the Depth configuration on seeded oracle modules (``finetune_unet`` as LdmDiffusers takes it)."""
    ldm = GradLdm(vae, unet, __import__("oracle.sd_modules", fromlist=["x"]).DDPMScheduler(), madm_path.uncond_stand_in(),
                  encoder_block_indices=[], unet_block_indices=[5, 8, 11], decoder_block_indices=(), input_range='-1+1',
                  unet_block_indices_type='after', vae_decoder_loss=cfg["vae_decoder_loss"])
    ldm.finetune_unet = finetune_unet
    ldm._freeze()                                                            # LdmDiffusers.__init__ (ldm_diffusers.py:77)
    gen = madm_path.PromptTimeGenerator(ldm)
    backbone = TrainBackbone(gen, cfg, in_size=in_size)
    head = madm_path.OracleHead(cfg)
    head.dropout = FixedDropout2d()
    crit = (criterion_cls or CmdiseCriterionRestated)(loss_weight=1.0)
    if train_palette is None:
        train_palette = [int(v) for v in torch.randint(0, 256, (cfg["num_classes"] * 3,), generator=torch.Generator().manual_seed(99))]
    return OracleMTMADISE(backbone, head, crit, train_palette=train_palette, **kw)
