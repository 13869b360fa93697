"""``MTMADISE`` -- the meta-architecture of the path with BOTH branches of the reference's ``forward``
(/root/reference/modeling/meta_arch/mtmadise.py:177-691 on top of CMDISE, modeling/meta_arch/cmdise.py:114-349):

* eval  (:657-691): ``MadmInference.forward`` (meta_arch.py);
* train (:180-656, BASELINE config 4): ``model(list[dict]) -> dict[str, scalar Tensor]`` as engine/train_loop.py:277-302
  expects it -- differentiable through standard autograd w.r.t. every ``requires_grad`` parameter, so
  ``scaler.scale(sum(losses.values())).backward()`` / DDP-style hooks / ``clip_grad_norm_`` work unchanged.

One training call = EMA update of the teacher (:184-185), source pass ('default' adapter, :239-256), ClassMix image
(:261-279), target pass on the mixed image (:284-302), teacher pass under no_grad with the EMA projections / head /
prompt (:308-337), pseudo labels + label / weight mixing (:339-392), the colour-label latents of the VAE-decoder losses
(:253-254, 394-397) and ``CmdiseCriterion`` (:441-492).  Every tensor op runs on libmadm_hip kernels: the two gradient
passes are recorded (block-boundary tape of the UNet, raw conv outputs of projections / head) and the losses come back as
the outputs of ONE autograd node (``_TrainStepFn``) whose backward walks those records with the explicit gradient kernels
(backward.py, head.py, backbone.py, criterion.py), weighting each loss by the upstream gradient it receives.

Options of the reference that no shipped config enables (mic / mic_reg, mask_diff, noise_reg, denoise_supervise, fd,
fd_attention, merge_with_pl_data, remove_amp / remove_texture, sem_seg_head_sec_modal=True, prompt masking) raise
NotImplementedError.  ``reg_uncertain`` only feeds the reference's visualisation (:323-328 -> vis_data :556-560); it is
accepted and changes nothing but the teacher call's ``return_unet_final_output`` (which the VAE-decoder branch computes
anyway).  The periodic matplotlib dump (``vis_results``) is out of scope.
"""
import os
import random
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from . import ops, labels as L, optim
from . import backward as bw
from .meta_arch import MadmInference
from .nn import Tok


class MTMADISE(MadmInference):
    psweight_ignore_top = 15          # cmdise.py:117

    def __init__(self, backbone, sem_seg_head, criterion, *, target_modality, train_palette, lora_configs=(),
                 vae_decoder_loss='', vae_decoder_loss_type=None, vae_decoder_loss_weight=(1.0, 1.0), reg_uncertain=False,
                 reg_target_palette=None, add_zero_grad=False, rev_noise_sup=False, rev_noise_end_iter=None,
                 rev_noise_gradually=False, denoise_timestep_range=None, eval_with_noise=None, max_iter=None,
                 sem_seg_head_sec_modal=False, ema_alpha=0.999, pseudo_threshold=0.968, blur=True, color_jitter_strength=0.2,
                 color_jitter_probability=0.2, enable_mixup=True, pl_crop=False, color_aug_flag=True, ema_w_unet=False,
                 pixel_mean=(0.0, 0.0, 0.0), pixel_std=(255.0, 255.0, 255.0), size_divisibility=64, color_aug=None,
                 **unsupported):
        for k, v in unsupported.items():
            if k in ("mic", "mic_reg", "mask_diff", "noise_reg", "denoise_supervise", "fd", "fd_attention",
                     "merge_with_pl_data", "remove_amp", "remove_texture", "prompt_confidence", "MIC_reg_wo_pl_val",
                     "w_rgb_lora", "wo_lora") and v:
                raise NotImplementedError(f"MTMADISE option {k}={v!r}: no shipped config enables it (SURVEY.md App. C.11)")
        if sem_seg_head_sec_modal:
            raise NotImplementedError("sem_seg_head_sec_modal=True (a second head copy) is not built")
        super().__init__(backbone, sem_seg_head, target_modality=target_modality, lora_configs=lora_configs,
                         pixel_mean=pixel_mean, pixel_std=pixel_std, size_divisibility=size_divisibility,
                         eval_with_noise=eval_with_noise)
        self.criterion = criterion
        self.train_iter_index = 0
        self.ema_alpha, self.pseudo_threshold = ema_alpha, pseudo_threshold
        # The EMA teacher's forward on a side stream beside the student's source pass: OFF by default since round 6.  Made race-free
        # (operands the teacher builds lazily on the side stream are ordered in front of main one by one: ops.side_builds), the
        # student trails the teacher layer by layer and the step is 0.4 .. 1.4 % SLOWER than in line (same box, three runs each:
        # 228.8 / 227.9 / 234.1 vs 225.7 / 224.4 / 233.1 ms, profiles/round6_ab_train_teacher_stream.txt); round 5's +1.5 % was
        # measured with the race in place.  MADM_TEACHER_OVERLAP=1 switches it on (tested: tests/test_train_gpu.py).
        self.overlap_teacher = bool(int(os.environ.get("MADM_TEACHER_OVERLAP", "0"))) and \
            not bool(int(os.environ.get("MADM_NO_TEACHER_OVERLAP", "0")))
        self._teacher_stream, self._teacher_warm = None, set()
        self.blur, self.color_jitter_strength, self.color_jitter_probability = blur, color_jitter_strength, color_jitter_probability
        self.enable_mixup, self.pl_crop, self.color_aug_flag = enable_mixup, pl_crop, color_aug_flag
        self.color_aug = color_aug        # callable(strong_parameters, data [N,3,H,W]) -> data; None: augment.strong_color
        self.vae_decoder_loss = vae_decoder_loss or ''
        self.vae_decoder_loss_type = vae_decoder_loss_type
        self.vae_decoder_loss_weight = list(vae_decoder_loss_weight)
        self.reg_uncertain, self.add_zero_grad = reg_uncertain, add_zero_grad
        self.rev_noise_sup, self.rev_noise_end_iter, self.rev_noise_gradually = rev_noise_sup, rev_noise_end_iter, rev_noise_gradually
        self.denoise_timestep_range = denoise_timestep_range
        self.train_max_iter = max_iter
        self.ema_w_unet = ema_w_unet
        self.train_palette = L.pad_palette(train_palette)                       # mtmadise.py:97-99
        if reg_target_palette is None:
            self.reg_target_palette = list(self.train_palette)
        else:
            assert reg_target_palette == 'discrete'
            self.reg_target_palette = L.pad_palette([255, 0, 255, 0, 255, 0, 127, 255, 127, 255, 127, 127, 0, 255, 255, 255,
                                                     255, 0, 0, 0, 255, 255, 0, 0, 127, 0, 127, 255, 255, 255, 0, 0, 0])
        self._inti_ema_weights()

    # ------------------------------------------------------------------ EMA teacher (cmdise.py:307-349)
    def _inti_ema_weights(self):
        bb = self.backbone
        bb.ema_feature_projections = deepcopy(bb.feature_projections)
        self.ema_sem_seg_head = deepcopy(self.sem_seg_head)
        self.ema_parms = [bb.ema_feature_projections, self.ema_sem_seg_head]
        self.updated_parms = [bb.feature_projections, self.sem_seg_head]
        ldm = bb.feature_extractor.ldm_extractor
        if self.ema_w_unet:
            ldm.ema_unet = deepcopy(ldm.unet)
            for m in ldm.ema_unet.modules():      # an EMA copy moves every step: no composed proj_out operand (sd_unet)
                if m.__class__.__name__ == "Transformer2DModel":
                    m.__dict__["weights_move"] = True
            self.ema_parms.append(ldm.ema_unet)
            self.updated_parms.append(ldm.unet)
        fe = bb.feature_extractor
        fe.ema_clip_project_others = deepcopy(fe.clip_project_others)
        self.ema_parms.append(fe.ema_clip_project_others)
        self.updated_parms.append(fe.clip_project_others)
        for m in self.ema_parms:
            for p in m.parameters():
                p.detach_()
                p.requires_grad = False
        # teacher parameters live in one flat buffer; student runs that are contiguous in an optimizer's flat buffer are
        # matched span-wise by optim.EmaPairs (a handful of launches per update instead of ~120 tiny ones)
        tp = [p for m in self.ema_parms for p in m.parameters()]
        sp = [p for m in self.updated_parms for p in m.parameters()]
        assert len(tp) == len(sp)
        self._ema_teacher_flat = optim.FlatParams(tp, with_grad=False) if tp and tp[0].is_cuda else None
        self._ema_pairs = optim.EmaPairs(tp, sp)

    def _update_ema(self, it):
        alpha_teacher = min(1 - 1 / (it + 1), self.ema_alpha)
        self._ema_pairs.update(alpha_teacher)

    # ------------------------------------------------------------------ helpers
    def _images(self, batched_inputs, key):
        """(x - pixel_mean) / pixel_std + ImageList zero padding to a multiple of 64 (:189-191,199-200)."""
        dev = next(self.parameters()).device
        xs = [x[key].to(dev).float() for x in batched_inputs]
        H, W = xs[0].shape[-2:]
        assert all(tuple(x.shape[-2:]) == (H, W) for x in xs), "training crops share one size (dataset mapper)"
        d = self.size_divisibility
        Hp, Wp = (H + d - 1) // d * d, (W + d - 1) // d * d
        return ops.scale_pad_nchw(torch.stack(xs).contiguous(), 1.0 / self.pixel_std, Hp, Wp)

    def _labels(self, batched_inputs):
        dev = next(self.parameters()).device
        g = torch.stack([x['source_label'].to(dev) for x in batched_inputs]).long()
        if g.dim() == 3:
            g = g[:, None]
        H, W = g.shape[-2:]
        d = self.size_divisibility
        assert H % d == 0 and W % d == 0, "label padding (ImageList pads labels with 0) is not needed by the shipped 512 crops"
        return g.contiguous()

    def _student_pass(self, img, input_modal):
        """backbone(img, return_unet_final_output=True, input_modal=...) + sem_seg_head, recorded for the backward:
        feature_extractor.py:156-170 -> ldm_base.py:832-924 -> ldm_diffusers.py:143-217 -> feature_extractor.py:367-396 ->
        daformer_head.py:702-749.  Returns (logit tokens Tok, unet_final_output dict, record)."""
        bb = self.backbone
        gen = bb.feature_extractor
        ldm = gen.ldm_extractor
        img = bb.preprocess_image(img)
        B = img.shape[0]
        project = gen.clip_project_rgb if input_modal == 'rgb' else gen.clip_project_others
        with torch.no_grad():
            cond_inputs, cond_emb = project(gen.uncond_inputs, None, repeat=B)
            batched = dict(img=img, cond_inputs=cond_inputs, cond_emb=cond_emb)
            st = ldm._stage_encode(batched)
            feats, extra = ldm._stage_unet(st, batched, _keep_for_grad=True, _return_tokens=True,
                                           return_unet_final_output=True)
            keep = ldm._grad_keep
            ldm._grad_keep = None
            proj_tape = []
            fd = bb.forward_features_recorded(feats, img.shape[-2:], proj_tape)
            head = self.sem_seg_head
            head_tape = {}
            toks = [fd['output_features'].tok[k] for k in head.in_keys]
            logits = head.forward_tokens(toks, tape=head_tape)
        no_grad_w = ()
        if ldm.vae_decoder_loss and not ldm.final_fuse_vae_decoder_feat:
            no_grad_w = (feats[0].W,)          # the decoder image is ``.detach()``ed (ldm_diffusers.py:196-201)
        rec = dict(keep=keep, proj_tape=proj_tape, head_tape=head_tape, project=project, B=B, no_grad_w=no_grad_w,
                   adapter=self.active_lora_adapter())     # the backward recomputes the fused LoRA operands: same adapter
        return logits, extra, rec

    def _color_latent(self, label, palette):
        """convert_label_to_rgb (:159-175) + vae_encoder(...)[0] (:253-254, 394-396)."""
        from .ldm_rocm import vae_encoder
        rgb, valid = L.convert_label_to_rgb(label, palette)
        lat = vae_encoder(vae=self.backbone.feature_extractor.ldm_extractor.vae, images=rgb, encoder_block_indices=[])[0]
        return lat, valid

    def _strong_color(self, strong_parameters, data):
        if not self.color_aug_flag:
            return data
        if self.color_aug is not None:
            return self.color_aug(strong_parameters, data)
        from . import augment
        return augment.strong_color(strong_parameters, data)

    # ------------------------------------------------------------------ forward
    def forward(self, batched_inputs):
        if not self.training:
            return MadmInference.forward(self, batched_inputs)
        return self.forward_train(batched_inputs)

    def forward_train(self, batched_inputs):
        with ops.sync_profile():       # eager launches, one step in flight: the lone-launch rows of the tile table
            return self._forward_train(batched_inputs)

    def _forward_train(self, batched_inputs):
        if self.train_iter_index > 0:
            self._update_ema(self.train_iter_index)
        with torch.no_grad():
            source = self._images(batched_inputs, 'source_rgb')
            assert not isinstance(batched_inputs[0]['target_second_modality'], dict)
            target = self._images(batched_inputs, 'target_second_modality')
            gt = self._labels(batched_inputs)                                             # [B, 1, H, W] i64
        B = source.shape[0]
        tmod = self.target_modality
        strong_parameters = {
            'mix': None,
            'color_jitter': random.uniform(0, 1),
            'color_jitter_s': self.color_jitter_strength,
            'color_jitter_p': self.color_jitter_probability,
            'blur': random.uniform(0, 1) if self.blur else 0,
            'mean': None, 'std': None,          # pixel_mean == 0: aug_mean / aug_std are None (cmdise.py:236-238)
        }
        crit = self.criterion
        K = self.sem_seg_head.num_classes
        ctxs = {}

        # ---- teacher: pseudo labels ----
        # The EMA teacher's forward reads nothing the student passes write (its own UNet / head copies, the frozen VAE) and its
        # result is needed only where the labels are mixed: it runs on a side stream beside the two student forwards (from the
        # second step on: the first builds the shared packed operands of the frozen modules in one stream).  Same kernels,
        # same values; the Python-side random draws keep their order (random: two uniforms above, this randint; numpy: the
        # class choices below).  MADM_NO_TEACHER_OVERLAP=1 for A/B runs.
        main_stream = torch.cuda.current_stream(source.device)
        # ... per input geometry: a batch / crop size seen for the first time (a short last batch) runs in line, so whatever a
        # first pass at that size builds lazily is built in ONE stream; the shared constant caches are stream-safe on top of
        # that (ldm_rocm._StreamSafeCache: an entry filled on the side stream is waited for by its users on other streams)
        warm_key = (B, tuple(source.shape[-2:]), tuple(target.shape[-2:]))
        overlap = self.overlap_teacher and warm_key in self._teacher_warm
        side = main_stream
        if overlap:
            if self._teacher_stream is None:
                self._teacher_stream = torch.cuda.Stream(device=source.device)
            side = self._teacher_stream
            side.wait_stream(main_stream)                 # inputs, and this step's EMA update, are in front of it
        # (operands the teacher pass builds lazily on the side stream -- it shares the student's UNet unless ema_w_unet -- are
        # ordered in front of the main stream's passes one by one: ops.side_builds / ops.note_build)
        with torch.no_grad(), torch.cuda.stream(side), ops.side_builds(main_stream):
            self.set_lora_adapter(state=tmod)
            kw = dict(input_modal='others', ema_forward=True)
            if self.rev_noise_sup and self.train_iter_index <= self.rev_noise_end_iter:
                t_ = random.randint(self.denoise_timestep_range[0], self.denoise_timestep_range[1])
                if self.rev_noise_gradually:
                    t_ = int(t_ * (1 - self.train_iter_index / self.rev_noise_end_iter))
                kw['timestep'] = (t_, t_ + 1)
            if self.reg_uncertain:
                low_res_feats, _ = self.backbone(target, return_unet_final_output=True, **kw)
            else:
                low_res_feats = self.backbone(target, **kw)
            head_t = self.ema_sem_seg_head
            ema_logits = head_t.forward_tokens([low_res_feats['output_features'].tok[k] for k in head_t.in_keys])
            ema_nchw = ops.nhwc_to_nchw(ema_logits.t, B, K, ema_logits.H, ema_logits.W)
            pseudo_prob, pseudo_label, pseudo_weight = L.pseudo_labels(ema_nchw, target.shape[2:], self.pseudo_threshold)
            del low_res_feats, ema_logits
        self._teacher_warm.add(warm_key)

        # ---- source pass ('default' adapter, input_modal 'rgb') ----
        self.set_lora_adapter(state='default')
        source_logits, source_out, rec_s = self._student_pass(source, 'rgb')
        with torch.no_grad():
            if 's' in self.vae_decoder_loss:
                source_color_gt_latent, source_color_gt_mask = self._color_latent(gt, self.reg_target_palette)

            # ---- mixed image (ClassMix + colour augmentation) ----
            if self.enable_mixup:
                mix_classes = L.get_class_choices(gt)        # get_class_masks' RNG calls; the masks are re-derived per use
                mixed = []
                for i in range(B):
                    _, img_i, _ = L.class_mix(gt[i], mix_classes[i], source[i], target[i])
                    mixed.append(self._strong_color(strong_parameters, img_i[None]))
                mixed_img = torch.cat(mixed)
            else:
                mixed_img = self._strong_color(strong_parameters, target.clone())

        # ---- target pass on the mixed image ----
        self.set_lora_adapter(state=tmod)
        target_logits, target_out, rec_t = self._student_pass(mixed_img, 'mixed')

        # ---- label / weight mixing with the teacher's pseudo labels ----
        if overlap:
            main_stream.wait_stream(side)
            for t_side in (ema_nchw, pseudo_prob, pseudo_label, pseudo_weight):
                t_side.record_stream(main_stream)
        with torch.no_grad():
            if self.pl_crop:
                pseudo_weight[:, :self.psweight_ignore_top, :] = 0
            if self.enable_mixup:
                gt_pixel_weight = torch.ones_like(pseudo_weight)
                lbls, wts = [], []
                for i in range(B):
                    _, w_i, l_i = L.class_mix(gt[i], mix_classes[i], gt_pixel_weight[i][None], pseudo_weight[i][None],
                                              label1=pseudo_label[i][None])
                    lbls.append(l_i)
                    wts.append(w_i)
                mixed_lbl = torch.stack(lbls)                                    # [B, 1, H, W]
                mixed_seg_weight = torch.cat(wts)                                # [B, H, W]
            else:
                mixed_lbl = pseudo_label[:, None]
                mixed_seg_weight = pseudo_weight
            if 't' in self.vae_decoder_loss:
                target_color_gt_latent, target_color_gt_mask = self._color_latent(mixed_lbl, self.reg_target_palette)
                target_color_gt_mask = target_color_gt_mask * pseudo_weight[:, None]

            # ---- losses (criterion.py:155-254) ----
            losses = {}
            losses['source_loss'], ctxs['source_loss'] = crit.ce_forward(source_logits, K, gt[:, 0])
            losses['target_loss'], ctxs['target_loss'] = crit.ce_forward(target_logits, K, mixed_lbl[:, 0],
                                                                         pixel_weight=mixed_seg_weight)
            if 's' in self.vae_decoder_loss:
                losses['vae_decoder_source_loss'], ctxs['vae_decoder_source_loss'] = crit.decoder_loss_forward(
                    source_out['before_vae.decoder'], source_color_gt_latent, source_color_gt_mask,
                    self.vae_decoder_loss_weight[0], self.vae_decoder_loss_type)
            if 't' in self.vae_decoder_loss:
                losses['vae_decoder_target_loss'], ctxs['vae_decoder_target_loss'] = crit.decoder_loss_forward(
                    target_out['before_vae.decoder'], target_color_gt_latent, target_color_gt_mask,
                    self.vae_decoder_loss_weight[1], self.vae_decoder_loss_type)
        self.train_iter_index += 1

        names = list(losses.keys())
        params = [p for p in self.parameters() if p.requires_grad]
        zero_ids = set()
        if self.add_zero_grad:          # :654-655: ``sum(torch.sum(p)) * 0.`` over the LoRA tensors of the other adapters --
            losses['zero_grad'] = torch.zeros((), device=source.device)     # value 0, gradient 0 (not None) for each of them
            names.append('zero_grad')
            zero_ids = {id(p) for p in self.unused_lora_parameters(tmod) if p.requires_grad}
        state = dict(model=self, names=names, ctxs=ctxs, rec_s=rec_s, rec_t=rec_t, params=params, zero_ids=zero_ids)
        self.last_step = dict(mixed_img=mixed_img, mixed_lbl=mixed_lbl, mixed_seg_weight=mixed_seg_weight,
                              pseudo_label=pseudo_label, pseudo_weight=pseudo_weight, ema_logits=ema_nchw,
                              source_logits=source_logits, target_logits=target_logits)
        outs = _TrainStepFn.apply(state, len(names), *[losses[n] for n in names], *params)
        return dict(zip(names, outs))

    def unused_lora_parameters(self, used_modal):
        """The tensors ``add_zero_gead_on_unused_lora`` sums (mtmadise.py:149-157): every UNet parameter whose name holds
        'lora' but not the target modality -- the 'default' adapter included."""
        unet = self.backbone.feature_extractor.ldm_extractor.unet
        return [p for name, p in unet.named_parameters() if 'lora' in name and used_modal not in name]

    def add_zero_gead_on_unused_lora(self, used_modal):
        """The reference's own formulation through torch autograd (kept for callers outside ``forward_train``, which
        folds the term into its autograd node: value 0, zero gradients, the tensors count as touched for AdamW)."""
        loss = [torch.sum(p) for p in self.unused_lora_parameters(used_modal)]
        return sum(loss) * 0.

    # ------------------------------------------------------------------ backward of one recorded student pass
    def _backward_pass(self, rec, dlogits, dsample, add):
        """dlogits: tokens [M, k_tile] (compute dtype) or None; dsample: NCHW f32 gradient of 'before_vae.decoder' or None;
        ``add(param, grad)`` accumulates a parameter gradient (called in the order the backward produces them: head,
        projections, UNet up -> mid -> down -> conv_in, batched K/V and time-embedding tails, prompt / time gates)."""
        bb = self.backbone
        gen = bb.feature_extractor
        ldm = gen.ldm_extractor
        self.set_lora_adapter(rec["adapter"])
        keep = rec["keep"]
        state = keep["state"]
        unet, dtype = keep["unet"], keep["dtype"]
        dins = {}
        if dlogits is not None:
            head = self.sem_seg_head
            dfeats, hg = head.backward_tokens(rec["head_tape"], dlogits)
            hp = dict(head.named_parameters())
            for k_, v in hg.items():
                add(hp[k_], v)
            dins, pg = bb.backward_features(rec["proj_tape"], dict(zip(head.in_keys, dfeats)), no_grad_inputs=rec["no_grad_w"])
            bp = dict(bb.named_parameters())
            for k_, v in pg.items():
                add(bp[k_], v)
        dtaps = []
        for tk in state.tapped:
            d = dins.get(tk.W)
            if d is None:
                d = torch.zeros_like(tk.t)
            dtaps.append(d if d.shape[1] == tk.C else d[:, :tk.C].contiguous())
        ds = None
        if dsample is not None:
            ds = ops.nchw_to_nhwc(dsample.float().contiguous(), dtype, max(unet.conv_out.n_pad, 16 // dsample.new_empty(0, dtype=dtype).element_size()))
        up = dict(unet.named_parameters())
        base = any(p.requires_grad and ".lora_" not in n for n, p in up.items())

        def unet_grad(k_, v):
            p = up.get(k_)
            if p is not None and p.requires_grad:
                add(p, v)

        res = bw.unet_backward_from_state(state, dtaps, dsample=ds, base_grads=base, grad_cb=unet_grad)
        # prompt / time conditioning
        project = rec["project"]
        if any(p.requires_grad for p in project.parameters()):
            B = rec["B"]
            dctx = ops.rows_to_f32(res["ctx"])[:, :gen.uncond_inputs.shape[2]].reshape(B, -1, gen.uncond_inputs.shape[2])
            pg = project.backward(gen.uncond_inputs, dctx.contiguous(), res["cond_emb"])
            pp = dict(project.named_parameters())
            for k_, v in pg.items():
                add(pp[k_], v)


class _TrainStepFn(torch.autograd.Function):
    """The whole training forward as ONE autograd node: outputs = the loss scalars the HIP forward computed, inputs = every
    trainable parameter; backward = the explicit gradient kernels over the two recorded student passes, each loss weighted
    by the gradient that reaches it (sum of the losses, GradScaler scale, per-loss weights -- all stay device scalars)."""

    @staticmethod
    def forward(ctx, state, n_losses, *rest):
        ctx.state = state
        ctx.n_losses = n_losses
        return tuple(v.clone() for v in rest[:n_losses])

    @staticmethod
    def backward(ctx, *gouts):
        with ops.sync_profile():
            return _TrainStepFn._backward(ctx, *gouts)

    @staticmethod
    def _backward(ctx, *gouts):
        st = ctx.state
        model, names, ctxs, params = st["model"], st["names"], st["ctxs"], st["params"]
        crit = model.criterion
        g = {n: (go if go is not None else None) for n, go in zip(names, gouts)}
        acc = {}
        touched = set()
        early = set()
        sink = getattr(model, "grad_sink", None)     # train.MadmTrainer: gradients go straight into its flat buffer and
        passes = []                                  # finished spans are all-reduced while the backward still runs
        dtype = st["rec_s"]["keep"]["dtype"]
        for rec, ce_name, dec_name in ((st["rec_s"], 'source_loss', 'vae_decoder_source_loss'),
                                       (st["rec_t"], 'target_loss', 'vae_decoder_target_loss')):
            if (g.get(ce_name) is not None) or (dec_name in ctxs and g.get(dec_name) is not None):
                passes.append((rec, ce_name, dec_name))
        for i, (rec, ce_name, dec_name) in enumerate(passes):
            last = i == len(passes) - 1

            def add(p, v, last=last):
                v = v.reshape(p.shape)
                touched.add(id(p))
                if sink is not None:                 # straight into the trainer's flat buffer (batched adds); the LAST
                    if last:                         # pass marks the parameter's span as finished
                        early.discard(id(p))
                        sink.final(p, v)
                    else:
                        early.add(id(p))
                        sink.accumulate(p, v)
                    return
                cur = acc.get(id(p))
                acc[id(p)] = v if cur is None else cur + v

            dlogits = crit.ce_backward(ctxs[ce_name], g[ce_name], dtype) if g.get(ce_name) is not None else None
            dsample = None
            if dec_name in ctxs and g.get(dec_name) is not None:
                dsample = crit.decoder_loss_backward(ctxs[dec_name], g[dec_name])
            ops.GRAD_ZEROS.reset(dlogits.device if dlogits is not None else dsample.device)   # one memset per pass
            adapter_now = model.active_lora_adapter()
            try:
                model._backward_pass(rec, dlogits, dsample, add)
            finally:
                model.set_lora_adapter(adapter_now)
            if sink is not None:
                sink.flush()
        ops.GRAD_ZEROS.drop()
        if g.get('zero_grad') is not None:     # zero gradients of the other adapters' LoRA tensors: present, not None
            for p in params:
                if id(p) in st["zero_ids"] and id(p) not in touched:
                    touched.add(id(p))
                    if sink is None:
                        acc[id(p)] = torch.zeros_like(p)
                    else:
                        early.add(id(p))       # the flat gradient buffer is already zero: only mark the span finished
        st["rec_s"] = st["rec_t"] = None       # free the tapes
        model.last_grad_param_ids = touched    # torch.optim.AdamW skips parameters whose grad is None
        if sink is not None:
            for p in params:                   # gradients only the first pass produced: already added, now finished
                if id(p) in early:
                    sink.final(p, None)
            sink.backward_done()
            return (None, None) + (None,) * ctx.n_losses + (None,) * len(params)
        return (None, None) + (None,) * ctx.n_losses + tuple(acc.get(id(p)) for p in params)
