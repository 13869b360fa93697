"""SD-v1-4 ``AutoencoderKL`` on the HIP path (diffusers 0.25 module/parameter names).

Replaces the diffusers modules the reference drives from
modeling/meta_arch/ldm_diffusers.py:283-311 (``vae_encoder``) and :314-346 (``vae_decoder``).
Parameter tree == diffusers' AutoencoderKL (83,653,863 parameters).
"""
import torch
import torch.nn as nn

from . import ops, packing
from .nn import Tok, Conv2d, Linear, GroupNorm, Identity, _Packed
from .sd_unet import ResnetBlock2D, Downsample2D, Upsample2D, _FusedProj


class VaeAttention(nn.Module):
    """Single-head spatial attention of the VAE mid block: GroupNorm -> q,k,v (with bias) ->
    softmax(QK^T/sqrt(C)) V -> to_out -> + input."""

    def __init__(self, channels, eps=1e-6):
        super().__init__()
        self.channels = channels
        self.group_norm = GroupNorm(32, channels, eps=eps)
        self.to_q = Linear(channels, channels)
        self.to_k = Linear(channels, channels)
        self.to_v = Linear(channels, channels)
        self.to_out = nn.ModuleList([Linear(channels, channels), Identity()])

    GEMM_MIN_L = 1024   # from this sequence length on, GEMM tiles beat the flash kernel at d = 512

    def _fused(self, key, layers):
        f = self.__dict__.get(key)
        if f is None or any(a is not b for a, b in zip(f._layers, layers)):
            f = _FusedProj(layers)
            self.__dict__[key] = f
        return f

    def forward(self, x):
        C, B, L = self.channels, x.B, x.HW
        h = self.group_norm(x)
        dtype = h.t.dtype
        if L >= self.GEMM_MIN_L and L % ops.k_tile(dtype) == 0:
            o = self._attend_gemm(h.t, B, L)
        else:
            qkv = self._fused("_f_qkv", [self.to_q, self.to_k, self.to_v])(h.t)
            o = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, 1, L, L, C, C ** -0.5)
        st = ops.new_chsums(B, C, x.t.device)
        out = self.to_out[0](o, residual=x.t, stats=st, B=B)
        return Tok(out, B, x.H, x.W, st)

    def _attend_gemm(self, h, B, L):
        """softmax(Q K^T / sqrt(C)) V as three GEMMs per image on the conv/linear kernel:
        S = Q K^T (f32 logits), P = softmax rows, O = P V.  V^T is produced directly by a GEMM with the operands
        swapped (V^T = W_v h^T); its bias is added after P V, exact because the rows of P sum to 1."""
        C = self.channels
        dtype = h.dtype
        qk = self._fused("_f_qk", [self.to_q, self.to_k])(h)          # [B*L, 2C] incl. biases
        wv, bv = self.to_v.packed(dtype)                                 # [C, C], f32 [C]
        o = torch.empty((B * L, C), dtype=dtype, device=h.device)
        for b in range(B):
            rows = slice(b * L, (b + 1) * L)
            vT = ops.linear(wv, h[rows])                                 # [C, L] = W_v h_b^T
            s = ops.linear(qk[rows, :C], qk[rows, C:], out_f32=True)     # [L, L] f32 logits
            p = ops.softmax_rows(s, dtype, C ** -0.5)
            ops.linear(p, vT, bias=bv, out=o[rows])
        return o


class UNetMidBlock2D(nn.Module):
    def __init__(self, channels, eps=1e-6):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(channels, channels, temb_channels=None, eps=eps),
                                      ResnetBlock2D(channels, channels, temb_channels=None, eps=eps)])
        self.attentions = nn.ModuleList([VaeAttention(channels, eps)])

    def forward(self, x):
        x = self.resnets[0](x)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            x = resnet(attn(x))
        return x


class DownEncoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, add_downsample, num_layers=2):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels=None, eps=1e-6)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, padding=0)]) if add_downsample else None


class UpDecoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, add_upsample, num_layers=3):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels=None, eps=1e-6)
            for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None


class Encoder(nn.Module):
    def __init__(self, in_channels=3, out_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        boc = tuple(block_out_channels)
        self.conv_in = Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, ch in enumerate(boc):
            in_ch, out_ch = out_ch, ch
            self.down_blocks.append(DownEncoderBlock2D(in_ch, out_ch, i != len(boc) - 1, layers_per_block))
        self.mid_block = UNetMidBlock2D(boc[-1])
        self.conv_norm_out = GroupNorm(32, boc[-1], eps=1e-6)
        self.conv_act = Identity()
        self.conv_out = Conv2d(boc[-1], 2 * out_channels, 3, padding=1)


class Decoder(nn.Module):
    def __init__(self, in_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        boc = tuple(block_out_channels)
        self.conv_in = Conv2d(in_channels, boc[-1], 3, padding=1)
        self.mid_block = UNetMidBlock2D(boc[-1])
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i, ch in enumerate(rev):
            prev, out_ch = out_ch, ch
            self.up_blocks.append(UpDecoderBlock2D(prev, out_ch, i != len(rev) - 1, layers_per_block + 1))
        self.conv_norm_out = GroupNorm(32, boc[0], eps=1e-6)
        self.conv_act = Identity()
        self.conv_out = Conv2d(boc[0], out_channels, 3, padding=1)


class _Config:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class AutoencoderKL(_Packed):
    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=4):
        super().__init__()
        self.config = _Config(scaling_factor=0.18215, latent_channels=latent_channels)
        self.latent_channels = latent_channels
        self.encoder = Encoder(3, latent_channels, block_out_channels)
        self.decoder = Decoder(latent_channels, 3, block_out_channels)
        self.quant_conv = Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = Conv2d(latent_channels, latent_channels, 1)

    def _versions(self):
        ps = [self.encoder.conv_in.weight, self.encoder.conv_in.bias,
              self.encoder.conv_out.weight, self.encoder.conv_out.bias, self.quant_conv.weight, self.quant_conv.bias,
              self.decoder.conv_in.weight, self.decoder.conv_in.bias, self.post_quant_conv.weight,
              self.post_quant_conv.bias]
        return tuple((p._version, p.data_ptr()) for p in ps)

    # ---- encoder: vae_encoder (ldm_diffusers.py:283-311) up to ``moments`` ----
    def encode_moments(self, x, encoder_block_indices=()):
        """x: ldm_rocm.RawImage (f32 NCHW image + its normalisation: the stem kernel madm_stem_conv3x3 reads it directly), or
        a Tok of the normalised image as im2col rows of the 3x3 stem (ops.image_to_im2col3x3; stems other than the SD
        VAE's 3 -> 128).  Returns (moments Tok [.., 8] = quant_conv(encoder(x)), taps list[Tok]).  quant_conv (1x1, 8->8)
        is folded into encoder.conv_out's weights: W' = Wq W, b' = Wq b + bq (exact composition)."""
        enc = self.encoder
        taps, index = [], 0
        dtype = x.t.dtype if isinstance(x, Tok) else x.dtype
        if not isinstance(x, Tok) and enc.conv_in.out_channels != 128:   # (not the SD VAE: the GEMM form of the stem)
            x = Tok(ops.image_to_im2col3x3(x.images, x.dtype, ops.k_tile(x.dtype), x.mean, x.std, x.minmax), x.B, x.H, x.W)
        if not isinstance(x, Tok):
            def build_stem_direct():   # [128,3,3,3] -> f32 [k = (r*3+s)*3 + c][128]
                w = enc.conv_in.weight.detach().float().permute(2, 3, 1, 0).reshape(27, enc.conv_in.out_channels)
                return w.contiguous(), enc.conv_in.bias.detach().float().contiguous()

            wT, bs = self._cache_get(("stem_direct",), build_stem_direct)
            st = ops.new_chsums(x.B, wT.shape[1], x.images.device)
            h = Tok(ops.stem_conv3x3(x.images, wT, bs, x.dtype, x.mean, x.std, stats=st, minmax=x.minmax),
                    x.B, x.H, x.W, st)
        else:
            def build_stem():   # [128,3,3,3] -> [128][k = (r*3+s)*3 + c] padded to the K-tile
                w = enc.conv_in.weight.detach().float().permute(0, 2, 3, 1).reshape(enc.conv_in.out_channels, 27)
                return (packing.pack_linear_weight(w, dtype, ops.k_tile(dtype)),
                        enc.conv_in.bias.detach().float().contiguous())

            ws, bs = self._cache_get((dtype, "stem"), build_stem)
            st = ops.new_chsums(x.B, ws.shape[0], x.t.device)
            h = Tok(ops.linear(x.t, ws, bias=bs, alg_nk=(enc.conv_in.out_channels, 27), stats=st, B=x.B),
                    x.B, x.H, x.W, st)
        for blk in enc.down_blocks:
            for resnet in blk.resnets:
                h = resnet(h)
                index += 1
                if index in encoder_block_indices:
                    taps.append(h)
            if blk.downsamplers is not None:
                for d in blk.downsamplers:
                    h = d(h)
        h = enc.mid_block(h)
        def build():
            W = enc.conv_out.weight.detach().double()            # [8, 512, 3, 3]
            b = enc.conv_out.bias.detach().double()
            Wq = self.quant_conv.weight.detach().double()[:, :, 0, 0]  # [8, 8]
            bq = self.quant_conv.bias.detach().double()
            Wf = torch.einsum("oc,cikl->oikl", Wq, W).float()
            bf = (Wq @ b + bq).float().contiguous()
            return packing.pack_conv_weight(Wf, dtype, ops.k_tile(dtype)), bf

        wp, bp = self._cache_get((dtype, "enc_out"), build)
        gn = None
        if ops.can_fuse_groupnorm(h.H, h.W, 3, 1, 1, False, False):
            if h.stats is None:
                h.stats = ops.new_chsums(h.B, h.C, h.t.device)
                ops.groupnorm_stats(h.t, h.B, h.HW, h.stats)
            no = enc.conv_norm_out
            gn = ([h.stats], no.weight.detach(), no.bias.detach(), no.num_groups, no.eps, True)
        else:
            h = enc.conv_norm_out(h, silu=True)
        o = ops.conv2d(h.t, wp, h.B, h.H, h.W, N=wp.shape[0], KH=3, KW=3, pad_t=1, pad_l=1, bias=bp, gn=gn)
        return h.like(o), taps

    # ---- decoder: vae_decoder (ldm_diffusers.py:314-346) ----
    def decode(self, z, decoder_block_indices=(), output_final=True):
        """z: Tok of latents [B*h*w, >=4] (first 4 channels used; e.g. the UNet's ``sample``).  Returns
        (sample Tok [.., 4] whose first 3 channels are the image, or None; taps list[Tok]).
        ``1/scaling_factor`` (:319) is folded into post_quant_conv's weights; the 4-channel tensors travel in
        zeroed K-tile-wide buffers so the 1x1 / 3x3 convs read them without a repacking pass."""
        dec = self.decoder
        dtype = z.t.dtype
        kt = ops.k_tile(dtype)
        M = z.t.shape[0]

        def build():
            Wp_ = self.post_quant_conv.weight.detach().float() * (1.0 / self.config.scaling_factor)
            return (packing.pack_conv_weight(Wp_, dtype, kt), self.post_quant_conv.bias.detach().float().contiguous())

        wp, bp = self._cache_get((dtype, "post_quant"), build)
        zin = z.t
        if zin.shape[1] != kt:   # widen the 4-channel latent to one K-tile (zeros above channel 3)
            buf = torch.zeros((M, kt), dtype=dtype, device=zin.device)
            ops.copy_columns(zin, buf, 4)
            zin = buf
        pq = torch.zeros((M, kt), dtype=dtype, device=zin.device)
        ops.conv2d(zin, wp, z.B, z.H, z.W, N=4, bias=bp, out=pq[:, :4])
        h = dec.conv_in(Tok(pq, z.B, z.H, z.W))
        h = dec.mid_block(h)
        taps, index = [], 0
        for blk in dec.up_blocks:
            for resnet in blk.resnets:
                if index in decoder_block_indices:   # 0-based, BEFORE the resnet (:330-333)
                    taps.append(h)
                index += 1
                h = resnet(h)
            if blk.upsamplers is not None:
                for u in blk.upsamplers:
                    h = u(h)
        if not output_final:
            return None, taps
        return dec.conv_out(h, norm=dec.conv_norm_out, stats=False), taps
