"""``LdmRocm``: the MI355X-native replacement of ``LdmDiffusers``
(/root/reference/modeling/meta_arch/ldm_diffusers.py:17-243) and of its helper functions
``vae_encoder`` (:283-311), ``vae_decoder`` (:314-346), ``add_noise`` (:349-360) and ``diffusion_unet``
(:454-616): same names, arguments, return structure and error behaviour, computed by the HIP
kernels of libmadm_hip.  Swap ``L(LdmDiffusers)`` for ``L(LdmRocm)`` in
config_files/common/models/mtmadise_multi_lora.py:26-35.
"""
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops, weights
from .nn import Tok
from .sd_unet import UNet2DConditionModel
from .sd_vae import AutoencoderKL


class DDPMSchedule:
    """SD-v1-4 scheduler_config.json (scaled_linear, 0.00085 -> 0.012, 1000 steps); only the
    add_noise coefficients are needed (ldm_diffusers.py:359).  Constants built on the host in f32
    exactly as diffusers does."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.sqrt_ac = self.alphas_cumprod ** 0.5
        self.sqrt_1mac = (1 - self.alphas_cumprod) ** 0.5
        self._dev = {}

    def tables(self, device):
        key = str(device)
        if key not in self._dev:
            # (a blocking host -> device copy: complete when .to() returns, so safe for every stream of the process)
            self._dev[key] = (self.sqrt_ac.to(device).contiguous(), self.sqrt_1mac.to(device).contiguous())
        return self._dev[key]


def _compute_dtype(m):
    return getattr(m, "compute_dtype", torch.bfloat16)


class RawImage:
    """The f32 NCHW image with its normalisation, handed to sd_vae.AutoencoderKL.encode_moments: the stem kernel
    (madm_stem_conv3x3) normalises and convolves in one pass."""

    def __init__(self, images, dtype, mean, std, minmax):
        self.images, self.dtype, self.mean, self.std, self.minmax = images, dtype, mean, std, minmax
        self.B, self.H, self.W = images.shape[0], images.shape[2], images.shape[3]


def _img_tokens(images, dtype, mean=0.0, std=1.0, minmax=None):
    """The image as encode_moments takes it (normalised inside the stem kernel)."""
    assert images.shape[1] == 3, "the SD VAE encoder takes 3-channel images"
    return RawImage(images.float().contiguous(), dtype, mean, std, minmax)


@torch.no_grad()
def vae_encoder(vae, images, encoder_block_indices):
    """Reference signature (ldm_diffusers.py:283-311): images [B,3,H,W] in [-1,1] (NCHW) ->
    (latents [B,4,H/8,W/8] f32 = posterior.mean * scaling_factor, list of tap tensors NCHW)."""
    dtype = _compute_dtype(vae)
    ops.ARENA.reset(images.device)
    x = _img_tokens(images, dtype)
    moments, taps = vae.encode_moments(x, tuple(encoder_block_indices))
    B, h, w = moments.B, moments.H, moments.W
    dev = images.device
    one, zero = _unit_tables(dev)
    lat, _ = ops.latents_add_noise(moments.t, vae.config.scaling_factor, _zeros(4 * h * w, dev), one, zero,
                                   _zeros_i64(B, dev), B, h * w, ops.k_tile(dtype), h, w)
    assert len(encoder_block_indices) == len(taps)
    return lat, [t.nchw() for t in taps]


@torch.no_grad()
def vae_decoder(vae, latents, decoder_block_indices, output_final=False):
    """Reference signature (ldm_diffusers.py:314-346): NCHW latents [B,4,h,w] (scaled by scaling_factor, as
    vae_encoder returns them) -> (image [B,3,8h,8w] f32 or None, list of tap tensors NCHW, taken BEFORE
    the indexed up-block resnet)."""
    dtype = _compute_dtype(vae)
    ops.ARENA.reset(latents.device)
    B, C, h, w = latents.shape
    z = Tok(ops.nchw_to_nhwc(latents.float().contiguous(), dtype, ops.k_tile(dtype)), B, h, w)
    sample, taps = vae.decode(z, tuple(decoder_block_indices), output_final)
    return (None if sample is None else sample.nchw(3)), [t.nchw() for t in taps]


class _StreamSafeCache(dict):
    """Lazily built device constants shared by every stream of the process.  An entry is filled by kernels on whatever
    stream is current at the first use (e.g. the EMA teacher's side stream, mtmadise.forward_train); a LATER user on another
    stream must not read it before those kernels ran.  Each entry keeps the event recorded behind its fill until that event
    has completed; users on other streams wait for it (free once complete: one query).  ``fills`` counts the misses.
    Fills during a stream capture carry no event (the fill is a node of that graph; constants are built in the warm-up)."""
    fills = 0

    def get_or_build(self, key, build):
        ent = self.get(key)
        if ent is None:
            val = build()
            ev = st = None
            t0 = val[0] if isinstance(val, tuple) else val
            if t0.is_cuda and not torch.cuda.is_current_stream_capturing():
                st = torch.cuda.current_stream(t0.device)
                ev = torch.cuda.Event()
                ev.record(st)
            self[key] = ent = [val, ev, st]
            _StreamSafeCache.fills += 1
            return val
        val, ev, st = ent
        if ev is not None and not torch.cuda.is_current_stream_capturing():
            if ev.query():
                ent[1] = ent[2] = None
            else:
                cur = torch.cuda.current_stream(st.device)
                if cur != st:
                    cur.wait_event(ev)
        return val


_const_cache = _StreamSafeCache()


def _unit_tables(dev):
    return _const_cache.get_or_build(("unit", str(dev)), lambda: (torch.ones(1, device=dev), torch.zeros(1, device=dev)))


def _zeros(n, dev):
    return _const_cache.get_or_build(("z", n, str(dev)), lambda: torch.zeros(n, device=dev))


def _const_timesteps(t, n, dev):
    return _const_cache.get_or_build(("ts", int(t), n, str(dev)),
                                     lambda: torch.full((n,), int(t), dtype=torch.int64, device=dev))


def _zeros_i64(n, dev):
    return _const_cache.get_or_build(("zi", n, str(dev)), lambda: torch.zeros(n, dtype=torch.int64, device=dev))


def add_noise(noise_scheduler, latents, timesteps, shared_noise=None):
    """Reference signature (ldm_diffusers.py:349-360) on NCHW f32 latents (computed in f32)."""
    B, C, h, w = latents.shape
    assert C == 4
    if shared_noise is None:
        raise NotImplementedError("add_noise without shared_noise is never used by LdmDiffusers.forward (:162-163)")
    noise = _sized_noise(shared_noise, (h, w))
    sa, sn = noise_scheduler.tables(latents.device)
    tok = ops.nchw_to_nhwc(latents.float().contiguous(), torch.float32, 32)
    _, noisy = ops.latents_add_noise(tok, 1.0, noise.contiguous(), sa, sn,
                                     timesteps.to(latents.device).long().contiguous(), B, h * w, 32, h, w)
    return ops.nhwc_to_nchw(noisy, B, 4, h, w)


_noise_cache = _StreamSafeCache()


def _sized_noise(shared_noise, hw):
    """shared_noise resized (bicubic, align_corners=False) when the latent is not 64x64
    (ldm_diffusers.py:351-353).  A constant of the model: resized once per size and cached."""
    hw = tuple(int(v) for v in hw)
    if tuple(shared_noise.shape[2:]) == hw:
        return shared_noise
    return _noise_cache.get_or_build((shared_noise.data_ptr(), hw), lambda: torch.nn.functional.interpolate(
        shared_noise, size=hw, mode="bicubic", align_corners=False).contiguous())


def diffusion_unet(unet, sample, timestep, encoder_hidden_states, res_time_embedding, unet_block_indices,
                   unet_block_indices_type):
    """Reference signature (ldm_diffusers.py:454-616): NCHW ``sample`` [B,4,h,w], ``timestep`` int64 [B],
    ``encoder_hidden_states`` [B,77,768], ``res_time_embedding`` [B,1,1280]|[B,1280]|None ->
    (namespace(sample=[B,4,h,w]), list of tap tensors NCHW f32)."""
    dtype = _compute_dtype(unet)
    ops.ARENA.reset(sample.device)
    B, C, h, w = sample.shape
    x = Tok(ops.nchw_to_nhwc(sample.float().contiguous(), dtype, ops.k_tile(dtype)), B, h, w)
    out, taps = _unet_tokens(unet, x, timestep, encoder_hidden_states, res_time_embedding, unet_block_indices,
                             unet_block_indices_type)
    return SimpleNamespace(sample=out.nchw(unet.out_channels)), [_tap_nchw(t) for t in taps]


def _tap_nchw(t):
    if isinstance(t, tuple):  # 'in'-type tap: (hidden, skip) concatenated along channels
        a, b = t
        return ops.nhwc_to_nchw([a.t, b.t], a.B, [a.C, b.C], a.H, a.W)
    return t.nchw()


def _unet_inputs(x, timestep, encoder_hidden_states, res_time_embedding):
    """The reference's argument forms -> (int64 timesteps [B], prompt tokens [B*Lk, 768] of the compute dtype, Lk,
    f32 time-embedding residual [B, 1280] or None)."""
    dtype = x.t.dtype
    B = x.B
    if not torch.is_tensor(timestep):
        timestep = torch.tensor([timestep], dtype=torch.int64, device=x.t.device)
    timestep = timestep.to(torch.int64).reshape(-1).expand(B).contiguous()
    ehs = encoder_hidden_states
    assert ehs.dim() == 3 and ehs.shape[0] == B, ehs.shape
    Lk = ehs.shape[1]
    ctx = ops.cast_from_f32(ehs.float().contiguous().view(B * Lk, ehs.shape[2]), dtype)
    cond = None
    if res_time_embedding is not None:
        cond = res_time_embedding
        if cond.dim() == 3 and cond.shape[1] == 1:
            cond = cond[:, 0]
        cond = cond.float().contiguous()
    return timestep, ctx, Lk, cond


def _unet_tokens(unet, x, timestep, encoder_hidden_states, res_time_embedding, unet_block_indices,
                 unet_block_indices_type):
    timestep, ctx, Lk, cond = _unet_inputs(x, timestep, encoder_hidden_states, res_time_embedding)
    return unet(x, timestep, ctx, Lk, cond_emb=cond, unet_block_indices=tuple(unet_block_indices),
                unet_block_indices_type=unet_block_indices_type)


class LdmRocm(nn.Module):
    """Drop-in for ``LdmDiffusers``: same class attributes (consumed by BasePromptTimeGenerator,
    ldm_base.py:769-774,940-960), constructor arguments, ``forward`` contract and ``_freeze``.

    Extra keyword-only arguments: ``compute_dtype`` (torch.bfloat16 fast mode / torch.float32 exact
    parity mode), ``weights`` ('pretrained' reads the diffusers snapshot directory given as
    ``stable_diffusion_name_or_path``; 'synthetic' draws seeded parameters -- no SD checkpoint exists
    offline), ``seed``, ``check_input_range``."""

    latent_image_size = (64, 64)
    text_embed_shape = torch.Size([77, 768])
    unet_time_embed_out_features = 1280
    uncond_inputs_size = torch.Size([1, 77, 768])
    feature_size = (512, 512)
    feature_dims = [512, 512, 2560, 1920, 960, 640, 512, 512]
    feature_strides = [4, 8, 64, 32, 16, 8, 8, 4]
    num_groups = 8
    grouped_indices = [[0], [1], [2], [3], [4], [5], [6], [7]]
    timesteps = 0
    input_mean = 0.5
    input_std = 0.5

    def __init__(self, stable_diffusion_name_or_path, encoder_block_indices, unet_block_indices,
                 decoder_block_indices, input_range='01', unet_block_indices_type='in', finetune_unet='no',
                 concat_pixel_shuffle=False, add_latent_noise=-1, norm_latent_noise=False, vae_decoder_loss=False,
                 input_channel_plus=0, final_fuse_vae_decoder_feat=False, *, compute_dtype=torch.bfloat16,
                 weights='pretrained', seed=0, check_input_range=True, device='cuda'):
        super().__init__()
        self.stable_diffusion_name_or_path = os.path.expanduser(stable_diffusion_name_or_path or "")
        self.encoder_block_indices = encoder_block_indices
        self.unet_block_indices = unet_block_indices
        self.decoder_block_indices = decoder_block_indices
        self.input_range = input_range
        assert self.input_range in {'01', '-1+1'}
        self.unet_block_indices_type = unet_block_indices_type
        assert self.unet_block_indices_type in {'in', 'after'}
        self.finetune_unet = finetune_unet
        assert self.finetune_unet in {'no', 'all', 'attention', 'without cross-attention'}
        for flag, name in ((concat_pixel_shuffle, 'concat_pixel_shuffle'), (input_channel_plus != 0, 'input_channel_plus'),
                           (add_latent_noise != -1, 'add_latent_noise'), (norm_latent_noise, 'norm_latent_noise')):
            if flag:
                raise NotImplementedError(f"LdmRocm: option {name} of LdmDiffusers (no shipped config enables it, "
                                          "SURVEY.md Appendix C.11) is not built yet")
        self.concat_pixel_shuffle = concat_pixel_shuffle
        self.add_latent_noise = add_latent_noise
        self.norm_latent_noise = norm_latent_noise
        self.input_channel_plus = input_channel_plus
        self.compute_dtype = compute_dtype
        self.check_input_range = check_input_range
        self.deferred_range_probes = None     # a list: forward passes append their (min, max) probe instead of syncing on it

        self.vae = AutoencoderKL()
        self.unet = UNet2DConditionModel()
        if weights == 'pretrained':
            weights_mod = weights_loader
            weights_mod.load_diffusers_dir(self.vae, self.stable_diffusion_name_or_path, "vae")
            weights_mod.load_diffusers_dir(self.unet, self.stable_diffusion_name_or_path, "unet")
        elif weights == 'synthetic':
            weights_loader.synth_init_(self.vae, seed, "vae.")
            weights_loader.synth_init_(self.unet, seed, "unet.")
        else:
            raise ValueError(weights)
        self.vae.compute_dtype = compute_dtype
        self.unet.compute_dtype = compute_dtype
        self.noise_scheduler = DDPMSchedule()

        rng = torch.Generator().manual_seed(42)
        self.register_buffer("shared_noise", torch.randn(1, self.vae.latent_channels, *self.latent_image_size,
                                                         generator=rng).detach())
        self.register_buffer("uncond_inputs", self._get_uncond_inputs('').detach())
        self.register_buffer("_minmax_init", torch.tensor([float("inf"), float("-inf")]), persistent=False)
        self.vae_decoder_loss = vae_decoder_loss
        self.final_fuse_vae_decoder_feat = final_fuse_vae_decoder_feat
        self.to(device)
        self._freeze()

    # ------------------------------------------------------------------ reference surface
    def _freeze(self):
        super().train(mode=False)
        for p in self.parameters():
            p.requires_grad = False
        if self.finetune_unet != 'no':
            if self.finetune_unet == 'all':
                for p in self.unet.parameters():
                    p.requires_grad = True
            elif self.finetune_unet == 'without cross-attention':
                for name, p in self.unet.named_parameters():
                    if 'attentions' in name and 'attn2' in name:
                        if 'to_v' in name:
                            assert self.unet.state_dict()[name].shape[1] == 768
                    else:
                        p.requires_grad = True
            else:
                for name, p in self.unet.named_parameters():
                    if 'attentions' in name:
                        p.requires_grad = True
            self.exclude_unused_params()

    def exclude_unused_params(self):
        """ldm_diffusers.py:123-141 runs a dummy forward / backward from ``unet_features[-1]`` and freezes every UNet
        parameter that received no gradient (DDP runs with find_unused_parameters=False, config_files/common/train.py:12).
        The same set, derived from the graph instead of a probe run: in forward order everything AFTER the last tap --
        up-block resnets / attentions whose flat index lies behind it ('in' taps: the tapped resnet itself too, the tap is
        its concatenated INPUT, :372-375,411-414), an up block's upsampler unless a later block still feeds the tap, and
        always conv_norm_out / conv_out (tests/test_bridge.py checks this against the reference's procedure)."""
        if not len(self.unet_block_indices):
            unused = [self.unet.conv_norm_out, self.unet.conv_out]
        else:
            last = max(self.unet_block_indices)            # taps are appended in forward order: [-1] is the largest index
            after = self.unet_block_indices_type == 'after'
            unused = [self.unet.conv_norm_out, self.unet.conv_out]
            idx = 0
            blocks = list(self.unet.up_blocks)
            first_idx = []
            for blk in blocks:
                first_idx.append(idx)
                for i, r in enumerate(blk.resnets):
                    used = idx <= last if after else idx < last
                    if not used:
                        unused.append(r)
                        if blk.has_cross_attention:
                            unused.append(blk.attentions[i])
                    idx += 1
            for k, blk in enumerate(blocks):
                if blk.upsamplers is None:
                    continue
                nxt = first_idx[k + 1] if k + 1 < len(blocks) else idx
                feeds = nxt <= last   # 'after': the next block's first resnet is used; 'in': the tap is (at or behind) its input
                if k + 1 >= len(blocks) or not feeds:
                    unused.extend(blk.upsamplers)
        for m in unused:
            for p in m.parameters():
                p.requires_grad = False

    def _get_uncond_inputs(self, text):
        """ldm_diffusers.py:219-243 runs the CLIP text encoder on '' once at construction.  The text
        encoder is outside the hot path and its weights are not available offline: a precomputed
        ``uncond_inputs.pt`` next to the snapshot is used when present, otherwise a seeded stand-in
        of the prompt-embedding scale (ldm_base.py:653)."""
        p = os.path.join(self.stable_diffusion_name_or_path, "uncond_inputs.pt")
        if self.stable_diffusion_name_or_path and os.path.exists(p):
            t = torch.load(p, map_location="cpu").float()
            assert tuple(t.shape) == tuple(self.uncond_inputs_size)
            return t
        return 0.02 * torch.randn(*self.uncond_inputs_size, generator=torch.Generator().manual_seed(4242))

    def forward(self, batched_inputs, input_modal, **kwargs):
        # two stages with a narrow hand-over (the noisy latents), so that a serving loop can capture / schedule them
        # separately: madm_amd/pipeline.py runs the encoder stage of every batch on one stream and the UNet stages of
        # consecutive batches side by side on three
        want_grad = torch.is_grad_enabled() and self._wants_grad(batched_inputs, kwargs)
        with torch.no_grad(), ops.sync_profile():       # (one batch in flight: the lone-launch rows; a no-op inside the graph runners)
            st = self._stage_encode(batched_inputs)
            hook = self.__dict__.get("stage_hook")
            if hook is not None:          # pipeline.StagedInference: the stage boundary of a whole-model forward (capture / stream switch)
                hook("encoded")
            out = self._stage_unet(st, batched_inputs, _keep_for_grad=want_grad, **kwargs)
        if not want_grad:
            return out
        return self._attach_unet_grad(out, batched_inputs, kwargs)

    # ---- training: the UNet stage as one autograd node ("differentiable w.r.t. its requires_grad
    #      params through standard autograd") ----
    def _wants_grad(self, batched_inputs, kwargs):
        if kwargs.get('ema_forward') and hasattr(self, 'ema_unet'):
            return False
        if kwargs.get("_return_tokens", False) or batched_inputs.get("return_unet_feats"):
            return False   # the HIP backbone's token hand-over has no backward yet
        t = batched_inputs['cond_inputs'], batched_inputs['cond_emb']
        return any(x is not None and torch.is_tensor(x) and x.requires_grad for x in t) or \
            any(p.requires_grad for p in self.unet.parameters())

    def _attach_unet_grad(self, out, batched_inputs, kwargs):
        """Re-issues the UNet taps (and the final sample, when it was asked for) as outputs of ``_UNetTapsFn``: the
        values are the ones the forward just computed, the node's backward is ``backward.unet_backward``."""
        keep = self._grad_keep
        self._grad_keep = None
        extra = None
        if isinstance(out, tuple):
            feats, extra = out
        else:
            feats = out
        n_enc = keep["n_enc"]
        n_taps = len(keep["taps"])
        with_sample = isinstance(extra, dict)
        named = [(n, p) for n, p in self.unet.named_parameters() if p.requires_grad]
        keep["param_names"] = [n for n, _ in named]
        keep["with_sample"] = with_sample
        values = list(feats[n_enc:n_enc + n_taps]) + ([extra['before_vae.decoder']] if with_sample else [])
        outs = _UNetTapsFn.apply(keep, batched_inputs['cond_inputs'], batched_inputs['cond_emb'], len(values), *values,
                                 *[p for _, p in named])
        feats = list(feats)
        feats[n_enc:n_enc + n_taps] = outs[:n_taps]
        if with_sample:
            extra = dict(extra)
            extra['before_vae.decoder'] = outs[n_taps]
            return feats, extra
        return feats if extra is None else (feats, extra)

    def _stage_encode(self, batched_inputs):
        """normalise -> vae_encoder -> timestep draw -> add_noise (ldm_diffusers.py:143-163); returns the hand-over."""
        images = batched_inputs['img']
        dtype = self.compute_dtype
        dev = images.device
        B, _, H, W = images.shape
        ops.ARENA.reset(dev)
        mean, std = (self.input_mean, self.input_std) if self.input_range == '-1+1' else (0.0, 1.0)
        minmax = None
        if self.input_range == '-1+1':   # device-to-device reset: capture-safe (no host memcpy in the graph)
            minmax = torch.empty(2, device=dev)
            minmax.copy_(self._minmax_init)
        x = _img_tokens(images, dtype, mean, std, minmax)

        # latents (vae_encoder) + timesteps + add_noise, the last two fused into one kernel
        moments, enc_taps = self.vae.encode_moments(x, tuple(self.encoder_block_indices))
        assert len(self.encoder_block_indices) == len(enc_taps)
        if 'timestep' in batched_inputs.keys():
            low_timestep, high_timestep = batched_inputs['timestep'][0], batched_inputs['timestep'][1]
        else:
            low_timestep, high_timestep = 0, 1
        if high_timestep - low_timestep == 1:   # a one-value range (every shipped config): no draw, no RNG kernel in the graph
            timesteps = _const_timesteps(low_timestep, B, dev)
        else:
            timesteps = torch.randint(low=low_timestep, high=high_timestep, size=(B,), device=dev).long()
        h, w = moments.H, moments.W
        noise = _sized_noise(self.shared_noise, (h, w))
        sa, sn = self.noise_scheduler.tables(dev)
        latents, noisy = ops.latents_add_noise(moments.t, self.vae.config.scaling_factor, noise, sa, sn, timesteps,
                                               B, h * w, ops.k_tile(dtype), h, w)
        self.last_latents = latents
        self.last_minmax = minmax     # the stem's (min, max) probe of this batch: pipeline.DeferredRangeCheck reads it late
        return {"B": B, "h": h, "w": w, "noisy": noisy, "latents": latents, "timesteps": timesteps, "enc_taps": enc_taps,
                "minmax": minmax}

    def _stage_unet(self, st, batched_inputs, **kwargs):
        """diffusion_unet on the noisy latents + the feature hand-over (ldm_diffusers.py:165-217)."""
        dtype = self.compute_dtype
        B, h, w = st["B"], st["h"], st["w"]
        noisy, latents, timesteps, enc_taps, minmax = st["noisy"], st["latents"], st["timesteps"], st["enc_taps"], st["minmax"]
        text_prompt = batched_inputs['cond_inputs']
        res_time_embedding = batched_inputs['cond_emb']

        if 'ema_forward' in kwargs.keys() and kwargs['ema_forward'] and hasattr(self, 'ema_unet'):
            forward_unet = self.ema_unet
        else:
            forward_unet = self.unet
        if kwargs.pop("_keep_for_grad", False):
            # training: the same forward, run block by block with the block inputs kept for backward.unet_backward_from_state
            if self.unet_block_indices_type != "after":
                raise NotImplementedError("autograd through 'in'-type taps")
            from . import backward as bw
            x_tok = Tok(noisy, B, h, w)
            ts_, ctx_, Lk_, cond_ = _unet_inputs(x_tok, timesteps, text_prompt, res_time_embedding)
            sample, unet_taps, state = bw.unet_forward_recorded(forward_unet, x_tok, ts_, ctx_, Lk_,
                                                                tuple(self.unet_block_indices), cond_emb=cond_)
            self._grad_keep = {"unet": forward_unet, "state": state, "B": B, "taps": [t_.C for t_ in unet_taps],
                               "n_enc": len(enc_taps), "dtype": dtype}
        else:
            sample, unet_taps = _unet_tokens(forward_unet, Tok(noisy, B, h, w), timesteps, text_prompt,
                                             res_time_embedding, self.unet_block_indices, self.unet_block_indices_type)

        hook = self.__dict__.get("stage_hook")
        if hook is not None:              # the UNet is done: what follows (VAE decoder branch, projections, head) is the third stage
            hook("unet")
        # feature lists as channels-last Toks; converted to NCHW f32 at the API boundary unless the (HIP) backbone
        # asked for tokens (madm_amd.backbone passes _return_tokens=True)
        as_tok = bool(kwargs.get("_return_tokens", False))
        enc_tok, unet_tok, dec_tok = list(enc_taps), list(unet_taps), []
        decoder_output = None
        if self.vae_decoder_loss:   # :192-201
            dec, _ = self.vae.decode(sample, (), output_final=True)
            decoder_output = dec        # Tok [.., 4], image in the first 3 channels; no autograd here: "detached"
            if self.final_fuse_vae_decoder_feat:
                dec_tok = [dec]
            else:
                assert len(self.encoder_block_indices) == 0
                enc_tok = [dec]
        elif len(self.decoder_block_indices) != 0:
            # taps of the decoder run on the (scaled) latents, not on the UNet output (:204-205)
            lat_tok = Tok(ops.nchw_to_nhwc(latents, dtype, ops.k_tile(dtype)), B, h, w)
            _, dec_tok = self.vae.decode(lat_tok, tuple(self.decoder_block_indices), output_final=False)
        if getattr(self, "_grad_keep", None) is not None:
            self._grad_keep["n_enc"] = len(enc_tok)
        if minmax is not None and self.check_input_range and not torch.cuda.is_current_stream_capturing():
            if self.deferred_range_probes is not None:
                # a caller that synchronises later anyway (train.MadmTrainer at the end of its step) collects the probes and
                # asserts there: the same check (:147) without one host sync per forward pass
                self.deferred_range_probes.append(minmax)
            else:
                lo, hi = minmax.tolist()  # the reference's range assert (:147); one sync per call, like there
                assert -1 <= lo and hi <= 1
        self.last_sample = sample

        def out(t):
            if as_tok:
                if isinstance(t, tuple):
                    raise NotImplementedError("'in'-type taps are handed over as NCHW tensors only")
                return t
            if t is decoder_output:
                return t.nchw(3)
            return _tap_nchw(t)

        feats = [out(t) for t in (*enc_tok, *unet_tok, *dec_tok)]
        if "return_unet_feats" in batched_inputs.keys() and batched_inputs["return_unet_feats"]:
            return feats, [out(t) for t in unet_tok]
        elif "return_unet_final_output" in kwargs.keys() and kwargs["return_unet_final_output"]:
            return feats, {
                'before_vae.decoder': sample.nchw(self.unet.out_channels),
                # like the reference (:214) this needs the vae_decoder_loss branch
                'after_vae.decoder': ops.clamp_f32(decoder_output.nchw(3), -1.0, 1.0),
            }
        else:
            return feats


class _UNetTapsFn(torch.autograd.Function):
    """The UNet stage of LdmRocm as ONE autograd node: forward hands back the tap tensors the (no-grad) HIP forward
    already produced; backward runs ``backward.unet_backward`` (block interiors recomputed) and returns the gradients of
    the prompt tokens, the time-embedding residual and every trainable UNet parameter."""

    @staticmethod
    def forward(ctx, keep, cond_inputs, cond_emb, n_values, *rest):
        ctx.keep = keep
        ctx.cond_shape = None if cond_emb is None else tuple(cond_emb.shape)
        ctx.n_values = n_values
        ctx.save_for_backward(cond_inputs, *([] if cond_emb is None else [cond_emb]))
        return tuple(v.clone() for v in rest[:n_values])

    @staticmethod
    def backward(ctx, *gouts):
        with ops.sync_profile():          # eager launches of a synchronous training step: the lone-launch rows
            return _UNetTapsFn._backward(ctx, *gouts)

    @staticmethod
    def _backward(ctx, *gouts):
        from . import backward as bw
        k = ctx.keep
        unet, dtype, B = k["unet"], k["dtype"], k["B"]
        saved = ctx.saved_tensors
        cond_inputs = saved[0]
        cond_emb = saved[1] if len(saved) > 1 else None
        n_taps = len(k["taps"])

        def tokens(g, cpad=None):   # NCHW f32 gradient -> channels-last tokens of the compute dtype
            return ops.nchw_to_nhwc(g.float().contiguous(), dtype, cpad if cpad is not None else g.shape[1])

        dtaps = [tokens(gouts[i]) for i in range(n_taps)]   # autograd materialises zeros for unused outputs
        dsample = None
        if k["with_sample"] and gouts[n_taps] is not None:
            dsample = tokens(gouts[n_taps], cpad=max(unet.conv_out.n_pad, 16 // torch.empty(0, dtype=dtype).element_size()))
        names = k["param_names"]
        base = any(".lora_" not in n for n in names)
        res = bw.unet_backward_from_state(k.pop("state"), dtaps, dsample=dsample, base_grads=base)
        g_ctx = None
        if cond_inputs.requires_grad:
            g_ctx = ops.rows_to_f32(res["ctx"])[:, :cond_inputs.shape[2]].reshape(cond_inputs.shape).to(cond_inputs.dtype)
        g_cond = None
        if cond_emb is not None and cond_emb.requires_grad:
            g_cond = res["cond_emb"].reshape(ctx.cond_shape).to(cond_emb.dtype)
        params = dict(unet.named_parameters())
        g_params = []
        for n in names:
            g = res["grads"].get(n)
            g_params.append(None if g is None else g.reshape(params[n].shape).to(params[n].dtype))
        return (None, g_ctx, g_cond, None) + (None,) * ctx.n_values + tuple(g_params)


weights_loader = weights
