"""ORACLE (test infrastructure only -- never imported by the product path in madm_amd/).

CPU fp32 restatement, in plain torch, of the un-vendored third-party arithmetic under the reference's
hot path: diffusers==0.25.0 ``UNet2DConditionModel`` / ``AutoencoderKL`` / ``DDPMScheduler`` in their
Stable-Diffusion-v1-4 configuration and peft==0.10.0 LoRA ``Linear``
(reference pins: /root/reference/requirements.txt:2,14; call sites
modeling/meta_arch/ldm_diffusers.py:11-13,246-266,283-616 and modeling/meta_arch/mtmadise.py:115-147).
Those packages are not present in /root/reference nor installable here, so the block semantics follow
the published SD-v1-4 architecture and are pinned by:
  K1 parameter totals (UNet 859,520,964; VAE 83,653,863; LoRA 199,296*r)     tests/test_oracle.py
  K2 the twelve up-block tap shapes of modeling/backbone/feature_extractor.py:321-346
  K4/K5 noise schedule and seed-42 shared noise (ldm_diffusers.py:73-75,349-360)
and, in this container, by driving these modules with the reference's OWN duck-typed orchestration
functions loaded from /root/reference (oracle/ref_driver.py).  Parity status of the block arithmetic
itself: "parity unpinned" by the reference (it ships no tests); see DESIGN.md.

Attribute and parameter names are the diffusers 0.25 ones, so (a) the reference's
``diffusion_unet`` / ``vae_encoder`` / ``vae_decoder`` can drive these objects unchanged and
(b) state_dicts are interchangeable with the product's parameter containers.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- building blocks
class Timesteps(nn.Module):
    """diffusers Timesteps(num_channels=320, flip_sin_to_cos=True, downscale_freq_shift=0)."""

    def __init__(self, num_channels=320, flip_sin_to_cos=True, downscale_freq_shift=0.0):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.downscale_freq_shift)
        emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels=320, time_embed_dim=1280):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


class ResnetBlock2D(nn.Module):
    """GN -> SiLU -> conv3x3 (+ time row) -> GN -> SiLU -> (dropout 0) -> conv3x3, + (1x1) shortcut."""

    def __init__(self, in_channels, out_channels, temb_channels=1280, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.output_scale_factor = 1.0

    def forward(self, input_tensor, temb=None, *args, **kwargs):
        h = self.conv1(self.nonlinearity(self.norm1(input_tensor)))
        if self.time_emb_proj is not None and temb is not None:
            h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


class Downsample2D(nn.Module):
    def __init__(self, channels, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, hidden_states, *args, **kwargs):
        if self.padding == 0:
            hidden_states = F.pad(hidden_states, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(hidden_states)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None, *args, **kwargs):
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        return self.conv(hidden_states)


class LoraLinear(nn.Module):
    """peft 0.10 lora.Linear: out = base(x) + sum_active lora_B(lora_A(x)) * (alpha / r).
    ``_active_adapter`` is the attribute the reference flips (mtmadise.py:144-147)."""

    def __init__(self, base_layer):
        super().__init__()
        self.base_layer = base_layer
        self.lora_A = nn.ModuleDict()
        self.lora_B = nn.ModuleDict()
        self.scaling = {}
        self._active_adapter = []
        self._disable_adapters = False

    @property
    def in_features(self):
        return self.base_layer.in_features

    @property
    def out_features(self):
        return self.base_layer.out_features

    def update_layer(self, name, r, lora_alpha, generator=None):
        self.lora_A[name] = nn.Linear(self.in_features, r, bias=False)
        self.lora_B[name] = nn.Linear(r, self.out_features, bias=False)
        self.scaling[name] = lora_alpha / r
        with torch.no_grad():  # init_lora_weights="gaussian": A ~ N(0, 1/r), B = 0
            self.lora_A[name].weight.normal_(mean=0.0, std=1.0 / r, generator=generator)
            self.lora_B[name].weight.zero_()

    def forward(self, x):
        out = self.base_layer(x)
        if self._disable_adapters:
            return out
        active = self._active_adapter if isinstance(self._active_adapter, (list, tuple)) else [self._active_adapter]
        for name in active:
            if name in self.lora_A:
                out = out + self.lora_B[name](self.lora_A[name](x)) * self.scaling[name]
        return out


class Attention(nn.Module):
    """diffusers Attention with AttnProcessor2_0 semantics (softmax(QK^T / sqrt(d)) V), optionally with the
    VAE's GroupNorm + spatial reshape + residual (``_from_deprecated_attn_block``)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False, out_bias=True,
                 norm_num_groups=None, eps=1e-5, residual_connection=False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.residual_connection = residual_connection
        self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps, affine=True) if norm_num_groups else None
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])

    def forward(self, hidden_states, encoder_hidden_states=None, **kwargs):
        residual = hidden_states
        spatial = hidden_states.dim() == 4
        if spatial:
            b, c, hh, ww = hidden_states.shape
            hidden_states = hidden_states.view(b, c, hh * ww).transpose(1, 2)
        if self.group_norm is not None:
            hidden_states = self.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q, k, v = self.to_q(hidden_states), self.to_k(ctx), self.to_v(ctx)
        B, L, inner = q.shape
        d = inner // self.heads
        q = q.view(B, L, self.heads, d).transpose(1, 2)
        k = k.view(B, -1, self.heads, d).transpose(1, 2)
        v = v.view(B, -1, self.heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, L, inner)
        o = self.to_out[1](self.to_out[0](o))
        if spatial:
            o = o.transpose(-1, -2).reshape(b, c, hh, ww)
        if self.residual_connection:
            o = o + residual
        return o


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, hidden_states, *args, **kwargs):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)  # exact erf GELU


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, hidden_states, *args, **kwargs):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, heads, dim_head, bias=False)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head, bias=False)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, hidden_states, encoder_hidden_states=None, **kwargs):
        hidden_states = self.attn1(self.norm1(hidden_states)) + hidden_states
        hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states=encoder_hidden_states) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states
        return hidden_states


class Transformer2DModel(nn.Module):
    """GN(32, eps 1e-6) -> conv1x1 -> tokens -> BasicTransformerBlock -> conv1x1 -> + input."""

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim=768, norm_num_groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, hidden_states, encoder_hidden_states=None, cross_attention_kwargs=None, attention_mask=None,
                return_dict=True, **kwargs):
        b, c, hh, ww = hidden_states.shape
        residual = hidden_states
        hidden_states = self.proj_in(self.norm(hidden_states))
        inner = hidden_states.shape[1]
        hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(b, hh * ww, inner)
        for blk in self.transformer_blocks:
            hidden_states = blk(hidden_states, encoder_hidden_states=encoder_hidden_states)
        hidden_states = hidden_states.reshape(b, hh, ww, inner).permute(0, 3, 1, 2).contiguous()
        out = self.proj_out(hidden_states) + residual
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)


# ----------------------------------------------------------------------------- UNet blocks
class CrossAttnDownBlock2D(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, heads, add_downsample=True, num_layers=2):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList()
        for i in range(num_layers):
            self.resnets.append(ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels))
            self.attentions.append(Transformer2DModel(heads, out_channels // heads, out_channels))
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, padding=1)]) if add_downsample else None
        self.gradient_checkpointing = False

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs=None, **kwargs):
        output_states = ()
        for resnet, attn in zip(self.resnets, self.attentions):
            hidden_states = resnet(hidden_states, temb)
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states).sample
            output_states += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states += (hidden_states,)
        return hidden_states, output_states


class DownBlock2D(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, add_downsample=False, num_layers=2):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, padding=1)]) if add_downsample else None
        self.gradient_checkpointing = False

    def forward(self, hidden_states, temb=None, **kwargs):
        output_states = ()
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb)
            output_states += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states += (hidden_states,)
        return hidden_states, output_states


class UNetMidBlock2DCrossAttn(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, heads):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels, in_channels), ResnetBlock2D(in_channels, in_channels)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, in_channels // heads, in_channels)])
        self.gradient_checkpointing = False

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs=None, **kwargs):
        hidden_states = self.resnets[0](hidden_states, temb)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states).sample
            hidden_states = resnet(hidden_states, temb)
        return hidden_states


class UpBlock2D(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, add_upsample=True, num_layers=3):
        super().__init__()
        self.resnets = nn.ModuleList()
        for i in range(num_layers):
            res_skip = in_channels if i == num_layers - 1 else out_channels
            res_in = prev_output_channel if i == 0 else out_channels
            self.resnets.append(ResnetBlock2D(res_in + res_skip, out_channels))
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None
        self.gradient_checkpointing = False

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None, **kwargs):
        for resnet in self.resnets:
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = resnet(torch.cat([hidden_states, res], dim=1), temb)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states


class CrossAttnUpBlock2D(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, prev_output_channel, out_channels, heads, add_upsample=True, num_layers=3):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList()
        for i in range(num_layers):
            res_skip = in_channels if i == num_layers - 1 else out_channels
            res_in = prev_output_channel if i == 0 else out_channels
            self.resnets.append(ResnetBlock2D(res_in + res_skip, out_channels))
            self.attentions.append(Transformer2DModel(heads, out_channels // heads, out_channels))
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None
        self.gradient_checkpointing = False

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                cross_attention_kwargs=None, upsample_size=None, attention_mask=None, **kwargs):
        for resnet, attn in zip(self.resnets, self.attentions):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = resnet(torch.cat([hidden_states, res], dim=1), temb)
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states).sample
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states


class UNet2DConditionModel(nn.Module):
    """SD-v1-4 UNet: block_out_channels (320, 640, 1280, 1280), 2 layers per block, 8 heads,
    cross_attention_dim 768.  ``block_out_channels`` may be shrunk for fast structural tests."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), heads=8,
                 cross_attention_dim=768, layers_per_block=2):
        super().__init__()
        boc = tuple(block_out_channels)
        self.config = SimpleNamespace(center_input_sample=False, in_channels=in_channels, class_embed_type=None,
                                      block_out_channels=boc, attention_head_dim=heads,
                                      cross_attention_dim=cross_attention_dim)
        time_embed_dim = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_proj = Timesteps(boc[0], True, 0)
        self.time_embedding = TimestepEmbedding(boc[0], time_embed_dim)
        self.class_embedding = None
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, ch in enumerate(boc):
            in_ch, out_ch = out_ch, ch
            final = i == len(boc) - 1
            if not final:
                self.down_blocks.append(CrossAttnDownBlock2D(in_ch, out_ch, heads, add_downsample=True,
                                                             num_layers=layers_per_block))
            else:
                self.down_blocks.append(DownBlock2D(in_ch, out_ch, add_downsample=False, num_layers=layers_per_block))
        self.mid_block = UNetMidBlock2DCrossAttn(boc[-1], heads)
        self.num_upsamplers = 0
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i in range(len(rev)):
            prev = out_ch
            out_ch = rev[i]
            in_ch = rev[min(i + 1, len(rev) - 1)]
            final = i == len(rev) - 1
            if not final:
                self.num_upsamplers += 1
            if i == 0:
                self.up_blocks.append(UpBlock2D(in_ch, prev, out_ch, add_upsample=not final,
                                                num_layers=layers_per_block + 1))
            else:
                self.up_blocks.append(CrossAttnUpBlock2D(in_ch, prev, out_ch, heads, add_upsample=not final,
                                                         num_layers=layers_per_block + 1))
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    # peft-through-diffusers surface used by mtmadise.py:115-147
    def add_adapter(self, adapter_config, adapter_name="default", generator=None):
        targets = tuple(adapter_config.target_modules)
        for name, module in list(self.named_modules()):
            if isinstance(module, LoraLinear):
                if name.endswith(targets):
                    module.update_layer(adapter_name, adapter_config.r, adapter_config.lora_alpha, generator)
                continue
            if isinstance(module, nn.Linear) and name.endswith(targets) and ".lora_" not in name \
                    and not name.endswith("base_layer"):
                parent_name, _, child = name.rpartition(".")
                parent = self.get_submodule(parent_name)
                wrapped = LoraLinear(module)
                wrapped.update_layer(adapter_name, adapter_config.r, adapter_config.lora_alpha, generator)
                if isinstance(parent, nn.ModuleList):
                    parent[int(child)] = wrapped
                else:
                    setattr(parent, child, wrapped)

    def set_adapter(self, names):
        names = [names] if isinstance(names, str) else list(names)
        for m in self.modules():
            if isinstance(m, LoraLinear):
                m._active_adapter = names

    def forward(self, sample, timestep, encoder_hidden_states, **kwargs):
        """Plain diffusers forward (no taps) -- the reference uses its own ``diffusion_unet`` instead."""
        emb = self.time_embedding(self.time_proj(timestep.expand(sample.shape[0])).to(self.dtype))
        sample = self.conv_in(sample)
        res = (sample,)
        for blk in self.down_blocks:
            if blk.has_cross_attention:
                sample, r = blk(sample, emb, encoder_hidden_states=encoder_hidden_states)
            else:
                sample, r = blk(sample, emb)
            res += r
        sample = self.mid_block(sample, emb, encoder_hidden_states=encoder_hidden_states)
        for blk in self.up_blocks:
            r = res[-len(blk.resnets):]
            res = res[:-len(blk.resnets)]
            if blk.has_cross_attention:
                sample = blk(sample, r, emb, encoder_hidden_states=encoder_hidden_states)
            else:
                sample = blk(sample, r, emb)
        sample = self.conv_out(self.conv_act(self.conv_norm_out(sample)))
        return SimpleNamespace(sample=sample)


class LoraConfig:
    """Stand-in for peft.LoraConfig with the four fields mtmadise.py:118-124 sets."""

    def __init__(self, r=8, lora_alpha=8, init_lora_weights="gaussian", target_modules=("to_k", "to_q", "to_v", "to_out.0")):
        assert init_lora_weights == "gaussian"
        self.r, self.lora_alpha = r, lora_alpha
        self.init_lora_weights = init_lora_weights
        self.target_modules = list(target_modules)


# ----------------------------------------------------------------------------- VAE
class DownEncoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, add_downsample, num_layers=2):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels=None, eps=1e-6)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, padding=0)]) if add_downsample else None

    def forward(self, hidden_states, *args, **kwargs):
        for r in self.resnets:
            hidden_states = r(hidden_states, temb=None)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
        return hidden_states


class UpDecoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, add_upsample, num_layers=3):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels=None, eps=1e-6)
            for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, hidden_states, *args, **kwargs):
        for r in self.resnets:
            hidden_states = r(hidden_states, temb=None)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


class UNetMidBlock2D(nn.Module):
    """VAE mid block: resnet, single-head attention (d = channels) with GroupNorm + residual, resnet."""

    def __init__(self, channels, eps=1e-6):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(channels, channels, temb_channels=None, eps=eps),
                                      ResnetBlock2D(channels, channels, temb_channels=None, eps=eps)])
        self.attentions = nn.ModuleList([Attention(channels, None, heads=1, dim_head=channels, bias=True,
                                                   norm_num_groups=32, eps=eps, residual_connection=True)])

    def forward(self, hidden_states, temb=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            hidden_states = attn(hidden_states)
            hidden_states = resnet(hidden_states, temb)
        return hidden_states


class Encoder(nn.Module):
    def __init__(self, in_channels=3, out_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        boc = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, ch in enumerate(boc):
            in_ch, out_ch = out_ch, ch
            self.down_blocks.append(DownEncoderBlock2D(in_ch, out_ch, add_downsample=i != len(boc) - 1,
                                                       num_layers=layers_per_block))
        self.mid_block = UNetMidBlock2D(boc[-1])
        self.conv_norm_out = nn.GroupNorm(32, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[-1], 2 * out_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, in_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        boc = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, boc[-1], 3, padding=1)
        self.mid_block = UNetMidBlock2D(boc[-1])
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i, ch in enumerate(rev):
            prev, out_ch = out_ch, ch
            self.up_blocks.append(UpDecoderBlock2D(prev, out_ch, add_upsample=i != len(rev) - 1,
                                                   num_layers=layers_per_block + 1))
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    def forward(self, z):
        z = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            z = b(z)
        return self.conv_out(self.conv_act(self.conv_norm_out(z)))


class DiagonalGaussianDistribution:
    """Only ``mean`` is used by the reference (ldm_diffusers.py:304-308)."""

    def __init__(self, parameters):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)


class AutoencoderKL(nn.Module):
    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=4):
        super().__init__()
        self.config = SimpleNamespace(scaling_factor=0.18215, latent_channels=latent_channels)
        self.latent_channels = latent_channels
        self.encoder = Encoder(3, latent_channels, block_out_channels)
        self.decoder = Decoder(latent_channels, 3, block_out_channels)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)


# ----------------------------------------------------------------------------- scheduler
class DDPMScheduler:
    """SD-v1-4 scheduler_config: scaled_linear betas 0.00085 -> 0.012, 1000 steps; only add_noise is used
    (ldm_diffusers.py:359)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sa = ac[timesteps] ** 0.5
        sn = (1 - ac[timesteps]) ** 0.5
        while sa.dim() < original_samples.dim():
            sa = sa.unsqueeze(-1)
            sn = sn.unsqueeze(-1)
        return sa * original_samples + sn * noise
