"""Inference forward of the MADM meta-architecture on the HIP path:
``MTMADISE.forward`` eval branch, /root/reference/modeling/meta_arch/mtmadise.py:657-691 -- /255 normalisation,
pad to a multiple of 64, LoRA adapter of the target modality, backbone('others') -> segmentation head ->
bilinear upsampling to the padded input size -> crop -> ``[{'sem_seg': [1, K, H, W]}]``; plus the evaluator's
argmax (evaluation/d2_evaluator.py:106) as a device kernel.  The self-training (train) branch is mtmadise.MTMADISE.
LoRA contract (mtmadise.py:48-54,115-147): ``'name_rN_aM'`` configs -> one peft-shaped adapter each on to_q / to_k / to_v /
to_out.0 of every UNet attention, all active after construction, ``ldm_extractor._freeze()`` re-applied (which, with
``finetune_unet='no'``, freezes the adapters too -- the reference's own behaviour), one adapter selected per pass."""
import torch
import torch.nn as nn

from . import ops


class MadmInference(nn.Module):
    def __init__(self, backbone, sem_seg_head, *, target_modality="Depth", lora_configs=(), pixel_mean=(0.0, 0.0, 0.0),
                 pixel_std=(255.0, 255.0, 255.0), size_divisibility=64, eval_with_noise=None):
        super().__init__()
        assert list(pixel_mean) == [0.0, 0.0, 0.0] and len(set(pixel_std)) == 1, \
            "the shipped configs normalise by /255 (mtmadise_multi_lora.py:83-84)"
        self.backbone = backbone
        self.sem_seg_head = sem_seg_head
        self.sem_seg_head_sec_modal = sem_seg_head       # sem_seg_head_sec_modal=False: the same head object
        self.target_modality = target_modality
        self.pixel_std = float(pixel_std[0])
        self.size_divisibility = size_divisibility
        self.eval_with_noise = eval_with_noise
        self.lora_configs = {}
        for cfg in lora_configs:                          # 'name_rN_aM' (mtmadise.py:48-54)
            name, rank, alpha = cfg.split('_')
            assert name in {'default', 'Infrared', 'Depth', 'Event'}
            self.lora_configs[name] = dict(rank=int(rank[1:]), alpha=int(alpha[1:]))
        if self.lora_configs:
            self.set_multi_lora()

    def set_multi_lora(self):
        from types import SimpleNamespace
        ldm = self.backbone.feature_extractor.ldm_extractor
        for name, c in self.lora_configs.items():
            ldm.unet.add_adapter(adapter_config=SimpleNamespace(r=c['rank'], lora_alpha=c['alpha'],
                                                                init_lora_weights="gaussian",
                                                                target_modules=["to_k", "to_q", "to_v", "to_out.0"]),
                                 adapter_name=name)
        ldm.unet.set_adapter(list(self.lora_configs.keys()))
        ldm._freeze()

    def set_lora_adapter(self, state):
        if len(self.lora_configs) == 0 or state is None:
            return
        if isinstance(state, str):
            state = [state]
        for module in self.backbone.feature_extractor.ldm_extractor.unet.modules():
            if hasattr(module, "_active_adapter"):
                module._active_adapter = state

    def active_lora_adapter(self):
        """The adapter list ``set_lora_adapter`` wrote last (None without LoRA): the explicit backward of a recorded pass
        re-derives the fused LoRA operands and must see the adapter that pass ran with."""
        if len(self.lora_configs) == 0:
            return None
        for module in self.backbone.feature_extractor.ldm_extractor.unet.modules():
            if hasattr(module, "_active_adapter"):
                return list(module._active_adapter)
        return None

    @torch.no_grad()
    def forward(self, batched_inputs):
        with ops.sync_profile():      # one image in flight: lone-launch rows (no-op inside GraphedInference / StagedInference)
            return self._forward(batched_inputs)

    def _forward(self, batched_inputs):
        assert len(batched_inputs) == 1
        assert 'modality_type' not in batched_inputs[0].keys()
        x = batched_inputs[0]['target_second_modality'].to(next(self.parameters()).device).float()
        ori_size = x.shape[1:]
        H, W = ori_size
        d = self.size_divisibility
        Hp, Wp = (H + d - 1) // d * d, (W + d - 1) // d * d
        # (x - 0) / 255 and ImageList zero padding in one kernel: NCHW f32 -> NCHW f32 via the token layout is
        # not needed; scale + pad is a plain device op of the layout kernels
        img = ops.scale_pad_nchw(x[None].contiguous(), 1.0 / self.pixel_std, Hp, Wp)
        self.set_lora_adapter(state=self.target_modality)
        kw = {'input_modal': 'others'}
        if self.eval_with_noise is not None:
            kw['timestep'] = (self.eval_with_noise, self.eval_with_noise + 1)
        feats = self.backbone(img, **kw)
        logits = self.sem_seg_head_sec_modal(feats)                   # [1, K, h0, w0]
        out = ops.resize_bilinear_nchw(logits, Hp, Wp)                # F.interpolate(bilinear, align_corners=False)
        if (Hp, Wp) != (H, W):
            out = ops.crop_nchw(out, H, W)
        return [{'sem_seg': out}]

    @staticmethod
    def predict_labels(output):
        """evaluation/d2_evaluator.py:106: ``output['sem_seg'].argmax(dim=0)`` (the reference indexes [0] first)."""
        return ops.argmax_nchw(output['sem_seg'].contiguous())[0]
