/*
 * madm_hip.h -- C ABI of libmadm_hip.so, the MI355X (gfx950) kernel library under the
 * SD-v1-4 single-timestep feature extractor of MADM.
 *
 * The reference (XiaRho/MADM) has no FFI of its own: its hot path is Python calling
 * third-party torch modules (diffusers 0.25 / peft 0.10).  Each entry point below names
 * the reference call site whose arithmetic it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated, freed
 *     or retained by the library; workspaces are sized by the *_workspace_bytes queries;
 *   - activations are channels-last: [B, H, W, C] == row-major [B*H*W, C] ("tokens");
 *   - weights are [N][K] row-major, K = KH*KW*(C1+C2) ordered (kh, kw, c), c running
 *     over source 1 then source 2 (the skip-concat of the reference's up blocks);
 *   - `dtype` selects the storage/MFMA type of activations and weights:
 *       MADM_F32  -> f32 storage, v_mfma_f32_16x16x4_f32   (exact-f32 parity mode)
 *       MADM_BF16 -> bf16 storage, v_mfma_f32_16x16x32_bf16 (fast mode); f32 accumulate;
 *       MADM_F16  -> fp16 storage, v_mfma_f32_16x16x32_f16 (same rate and kernels as bf16, 3 more mantissa
 *                    bits: the reference's own autocast arithmetic, engine/train_loop.py:277,
 *                    evaluation/evaluator.py:62-66); f32 accumulate;
 *     bias / time rows / norm parameters / statistics are always f32 (f64 for GN sums);
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous on it,
 *     capture-safe (no allocation, no synchronisation), re-entrant across streams;
 *   - return value: 0 = ok, negative = error (see madm_status); the message of the last
 *     error on the calling thread is returned by madm_last_error().
 */
#ifndef MADM_HIP_H
#define MADM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MADM_ABI_VERSION 5

typedef enum {
    MADM_OK = 0,
    MADM_ERR_INVALID_ARG = -1,
    MADM_ERR_UNSUPPORTED = -2,
    MADM_ERR_LAUNCH = -3
} madm_status;

typedef enum { MADM_F32 = 0, MADM_BF16 = 1, MADM_F16 = 2 } madm_dtype;

/* activation codes of the norm / conv-input fusions */
typedef enum { MADM_ACT_NONE = 0, MADM_ACT_SILU = 1, MADM_ACT_RELU = 2 } madm_act;

typedef enum {
    MADM_EPI_NONE = 0,
    /* weight rows interleaved (value_j, gate_j); out[m][j] = value * gelu_erf(gate); the
     * output has N/2 columns.  diffusers GEGLU inside BasicTransformerBlock.ff, reached
     * from ldm_diffusers.py:436-440,528-534,553-559 */
    MADM_EPI_GEGLU = 1,
    /* out = relu(acc + bias + ... + residual): mmcv ConvModule(conv -> BatchNorm (folded into w / bias in
     * eval mode) -> ReLU) of the DAFormer head (modeling/sem_seg_head/daformer_head.py:364-372,455-461) */
    MADM_EPI_RELU = 2
} madm_epilogue;

int madm_abi_version(void);
const char* madm_last_error(void);

/* Calibration loop (ABI 4; no reference counterpart: measurement infrastructure of bench.py's "calib" object): `blocks`
 * workgroups of four waves issue `iters` x 8 independent v_mfma_f32_16x16x32_f16 each, operands in registers.  *flop (host,
 * optional) receives the FLOP of the launch; time it with events on `stream`: 512 blocks on MI355X = two waves per SIMD on
 * every CU, the rate this DEVICE sustains under a chip-wide MFMA load.  sink: 4 bytes of device memory, never written. */
int madm_calib_mfma_loop(int iters, int blocks, float* sink, double* flop, void* stream);

/* ---------------------------------------------------------------------------------
 * madm_conv2d_fwd: implicit-GEMM convolution / linear layer on MFMA.
 *   out[m][n] = epi( sum_k A(m,k) * w[n][k] + bias[n] + rowvec[b(m)][n] ) * 1 + residual[m][n]
 * A(m,k) gathers the (optionally nearest-2x upsampled, optionally two-source concatenated)
 * channels-last input at output pixel m = (b, oy, ox) and tap k = (kh, kw, c); zero outside.
 * Replaces: every nn.Conv2d / nn.Linear executed by diffusers under
 *   vae_encoder            modeling/meta_arch/ldm_diffusers.py:283-311
 *   diffusion_unet         modeling/meta_arch/ldm_diffusers.py:454-616
 *   diffusion_upblock2d / diffusion_cross_attn_upblock2d  ldm_diffusers.py:363-451
 *     (torch.cat skip concat :370,:409 -> in2/C2; Upsample2D -> upsample=1)
 *   vae_decoder            modeling/meta_arch/ldm_diffusers.py:314-346
 * including ResnetBlock2D's "+ time_emb_proj(silu(temb))[:, :, None, None]" (rowvec) and
 * its output residual / 1x1 shortcut sum, and peft LoRA (mtmadise.py:115-147) through the
 * K-concatenation [x | s*(xA^T)] x [W | B]^T (in2/C2 with KH=KW=1).
 * ------------------------------------------------------------------------------- */
typedef struct {
    int dtype;            /* madm_dtype */
    const void* in1;      /* [B, IH, IW, C1] (pixel stride ld1); 16-byte aligned, < 2 GiB */
    const void* in2;      /* [B, IH, IW, C2] or NULL when C2 == 0 */
    int C1, C2;           /* multiples of the K-tile: 64 (bf16) / 32 (f32) elements */
    int ld1, ld2;         /* pixel (row) strides of the sources in elements; 0 = dense (C1 / C2) */
    int B, IH, IW;        /* source dims (before the optional 2x upsample) */
    int OH, OW;
    int KH, KW;           /* 1x1 or 3x3 (any odd size works) */
    int stride;
    int pad_t, pad_l;     /* bottom/right padding is implied by OH/OW (zero fill) */
    int upsample;         /* 1: nearest 2x upsample of the sources before the conv */
    const void* w;        /* [N][K] of dtype, K = KH*KW*(C1+C2) */
    int ldw;              /* row stride of w in elements; 0 = dense (K) */
    int N;                /* output channels; multiple of 4 */
    const float* bias;    /* [N] or NULL */
    const float* rowvec;  /* [B][ldrv] or NULL: per-image row added to every pixel */
    int ldrv;             /* row stride of rowvec in floats (>= N, multiple of 4) */
    const void* residual; /* [M][ldr] of dtype or NULL; added after the epilogue */
    int ldr;
    void* out;            /* [M][ldo] of dtype (f32 when out_f32), M = B*OH*OW */
    int ldo;
    int out_f32;          /* 1: store f32 whatever the compute dtype (attention logits of the GEMM-based
                           * VAE mid-block attention); plain epilogue, no residual */
    int epilogue;         /* madm_epilogue */
    double* stats;        /* NULL, or f64 [B][N][2], zeroed by the caller: receives the per-(image, channel)
                           * sum and sum of squares of the stored output -- the statistics pass of the
                           * GroupNorm that consumes this tensor, fused into the producer's epilogue */
    /* fused GroupNorm(+act) of the INPUT (3x3 / stride 1 / pad 1 convs on maps >= 8x16 only, see
     * madm_conv2d_can_fuse_groupnorm): the conv reads RAW sources, folds the per-channel sums of the sources
     * (gn_sums1: f64 [B][C1][2], gn_sums2: f64 [B][C2][2] for the second source; the "stats" output of the producing
     * convs or madm_groupnorm_stats) into group mean / rstd in its prologue -- no finalize launch -- and applies
     * y = act((x - mean_g) * rstd_g * gn_gamma[c] + gn_beta[c]) while staging its LDS halo tile, zero padding after
     * the activation.  gn_sums1 NULL = off; gn_groups <= 32 and divides C1 + C2; gn_gamma / gn_beta f32 [C1 + C2];
     * gn_act: madm_act (SiLU: ResnetBlock2D norm1/norm2 + nonlinearity, conv_norm_out + conv_act; ReLU: d2
     * BottleneckBlock conv1.norm + relu before its 3x3 conv2) */
    const double* gn_sums1;
    const double* gn_sums2;
    const float* gn_gamma;
    const float* gn_beta;
    int gn_groups;
    float gn_eps;
    int gn_act;
    int splitk;           /* >=1; >1 needs workspace (f32 [splitk][M][N]) */
    void* workspace;
    size_t workspace_bytes;
    /* LayerNorm of the INPUT rows folded into a linear layer (ABI 2; diffusers BasicTransformerBlock norm1 -> attn1
     * to_q/k/v, norm2 -> attn2.to_q, norm3 -> ff.net.0.proj, modeling/meta_arch/ldm_diffusers.py:454-616 runs them as
     * nn.LayerNorm + nn.Linear): in1 holds the RAW rows x [M][C1]; w holds W' = W * gamma (column k scaled by gamma[k]);
     * ln_colsum f32 [N] = sum_k W'[n][k] (of the ROUNDED W'); bias = W beta + b.  The kernel takes sum x / sum x^2 of
     * every row from the operand fragments it reads anyway and stores
     *     out[m][n] = rstd_m * (sum_k x[m][k] W'[n][k] - mean_m * ln_colsum[n]) + bias[n]    (then the epilogue)
     * == Linear(LayerNorm(x)); no separate normalisation pass, no re-read.  NULL = off.  Needs KH = KW = 1, one source,
     * splitk == 1 (every workgroup walks whole rows), no fused GroupNorm. */
    const float* ln_colsum;
    float ln_eps;
    /* GroupNorm(+act) of the OUTPUT applied by the split-K reduction (ABI 3; diffusers ResnetBlock2D conv1 -> norm2 ->
     * nonlinearity, ldm_diffusers.py runs them as nn.Conv2d + nn.GroupNorm + SiLU): with splitk > 1 the slabs are summed by
     * one workgroup per (image, group), which holds its HW x N / pn_groups f32 values in LDS, takes the group's mean /
     * variance from them and stores out = act((v - mean) * rstd * pn_gamma[n] + pn_beta[n]): the raw conv output never
     * reaches memory and the stand-alone reduction + GroupNorm passes become one launch.  pn_gamma NULL = off.  Needs the
     * effective splitk > 1, the plain epilogue, no residual / stats / out_f32, N / pn_groups even and the group's values
     * within 96 KB of LDS (madm_conv2d_can_post_groupnorm tells). */
    const float* pn_gamma;
    const float* pn_beta;
    int pn_groups;
    float pn_eps;
    int pn_act;           /* madm_act */
} madm_conv2d_args;

size_t madm_conv2d_workspace_bytes(const madm_conv2d_args* a);
/* heuristic split-K for the MI355X grid (256 CUs); returns 1 when the tile grid already fills it */
int madm_conv2d_suggest_splitk(const madm_conv2d_args* a);
int madm_conv2d_fwd(const madm_conv2d_args* a, void* stream);
/* 1 when these arguments can take gn_sums1 / gn_gamma / ... (the LDS halo-tile 3x3 kernel applies), else 0. */
int madm_conv2d_can_fuse_groupnorm(const madm_conv2d_args* a);
/* 1 when madm_conv2d_fwd with these arguments (splitk, dims, pn_groups set; pointers not needed) can take pn_gamma / ...,
 * else 0. */
int madm_conv2d_can_post_groupnorm(const madm_conv2d_args* a);
/* which kernel instance madm_conv2d_fwd will launch for these arguments: 1 = igemm 128x128,
 * 2 = igemm 128x64, 3 = igemm 64x64, 4 = halo conv3x3 x128 channels, 5 = halo conv3x3 x64 channels,
 * 6 = igemm 64x64 with the 8-deep prefetch, 7 / 8 = LDS-DMA igemm 64x64 / 128x64,
 * 9 / 10 = halo conv3x3 x128 / x64 with LDS-DMA weights, 11 = LDS-DMA igemm 64x64 with the short ring
 * (used by bench.py to attribute time). */
int madm_conv2d_pick_tile(const madm_conv2d_args* a);
/* 1 when a row of the tuned table (madm_amd/csrc/igemm_tuned.inc, MADM_TUNED_FILE) decides this launch's tile / split-K, 0 when
 * it falls through to the heuristics (tests/test_parity_gpu.py::test_bench_workloads_have_tuned_rows: a shape of the bench
 * workloads without a row ran 2 x too long for a whole round). */
int madm_conv2d_has_tuned_row(const madm_conv2d_args* a);
/* Tile / split-K table profile of the process (ABI 5): 0 = throughput -- the rows tuned with the layer's launches side by side on three
 * streams, the default and what the graph runners of madm_amd/pipeline.py capture under; 1 = latency -- rows tuned for one launch on an
 * idle chip are consulted first (a synchronous caller with one batch in flight: the reference's own loop,
 * engine/train_loop.py:257-311 / evaluation/evaluator.py:75-93).  Both produce the same values up to summation order. */
int madm_set_tuning_profile(int profile);
int madm_get_tuning_profile(void);
/* tuning/debug aid: force the workgroup tile (0 = tuned table then heuristic, -1 = heuristic only,
 * 1..11 = the tile codes of madm_conv2d_pick_tile). */
void madm_debug_set_conv_tile(int tile);
/* test aid: fills the LDS of every CU with `pattern` (0x7fc00000 = quiet NaN) -- a kernel that reads LDS it has not written then
 * fails deterministically instead of depending on its predecessor on the CU.  sink: 4 bytes of device memory, never written. */
int madm_debug_poison_lds(unsigned pattern, void* sink, void* stream);

/* ---------------------------------------------------------------------------------
 * GroupNorm (32 groups in SD-v1-4; any G dividing Ctot), channels-last.  The normalised tensor has
 * Ctot channels and may be the channel concatenation [source 1 | source 2] (the skip concat of the up
 * blocks, ldm_diffusers.py:370,409).  Statistics are per-(image, channel) f64 (sum, sum of squares)
 * arrays `chsums` [B][C][2], one per source: produced either by the epilogue of the conv that wrote
 * the source (madm_conv2d_args.stats) or by madm_groupnorm_stats.
 *   stats: adds x's channel sums into the caller-zeroed chsums[B][C][2].
 *   apply: y[.., c_off + c] = act((x - mean_g) * rstd_g * gamma[c_off + c] + beta[c_off + c] + residual) for
 *          the dense source x [B*HW][C] occupying channels [c_off, c_off + C) of the concatenation; group
 *          statistics come from sums1 (channels [0, C1)) and sums2 (channels [C1, Ctot), may be NULL
 *          when C1 == Ctot); y has row stride ldy; act is a madm_act; residual (may be NULL, row stride
 *          ldres, same channel window) is added before the activation: detectron2 BottleneckBlock's
 *          relu(norm(conv3) + shortcut) (modeling/backbone/feature_extractor.py:347-359).
 * Replaces diffusers ResnetBlock2D.norm1/norm2 + nonlinearity, Transformer2DModel.norm,
 * Attention.group_norm (VAE), conv_norm_out + conv_act
 * (ldm_diffusers.py:290,297,299-300,387,435,553,609-610).
 * ------------------------------------------------------------------------------- */
int madm_groupnorm_stats(int dtype, const void* x, int B, int HW, int C, double* chsums, void* stream);
int madm_groupnorm_apply(int dtype, const void* x, void* y, int ldy, int B, int HW, int C,
                         int c_off, int Ctot, int G, const double* sums1, int C1, const double* sums2,
                         const float* gamma, const float* beta, float eps, int act,
                         const void* residual, int ldres, void* stream);
/* both sources of a channel-concatenated input ([x1 | x2], UNet up-block resnets: torch.cat([hidden, skip]) then
 * ResnetBlock2D.norm1) in ONE launch: y[:, :C1] / y[:, C1:] = act(GroupNorm([x1 | x2])), groups over the concatenation. */
int madm_groupnorm_apply_cat(int dtype, const void* x1, const void* x2, void* y, int ldy, int B, int HW, int C1, int C2,
                             int G, const double* sums1, const double* sums2, const float* gamma, const float* beta,
                             float eps, int act, void* stream);

/* channel sums -> the per-(image, channel) affine of the GroupNorm, scale/shift f32 [B][Ctot]:
 * y = x * scale + shift == (x - mean_g) * rstd_g * gamma + beta (what the fused conv computes in its prologue). */
int madm_groupnorm_finalize(int B, int HW, int Ctot, int G, const double* sums1, int C1,
                            const double* sums2, const float* gamma, const float* beta, float eps,
                            float* scale, float* shift, void* stream);

/* LayerNorm over the last dim of [M][C] (BasicTransformerBlock.norm1/2/3, eps 1e-5). */
int madm_layernorm_fwd(int dtype, const void* x, void* y, int M, int C,
                       const float* gamma, const float* beta, float eps, void* stream);

/* ---------------------------------------------------------------------------------
 * madm_attention_fwd: softmax(Q K^T * scale) V, flash style (no L x L matrix in HBM).
 *   q: [B, Lq, H, D] addressed as q + (b*Lq + i)*ldq + h*D ; k, v likewise with Lk, ldk, ldv
 *   o: [B, Lq, H, D] with row stride ldo.
 * Self-attention (Lk = Lq in {4096,1024,256,64}, D in {40,80,160}), cross-attention over
 * the 77-token prompt (Lk = 77) and the VAE mid-block attention (H = 1, D = 512).
 * Replaces diffusers Attention / AttnProcessor2_0 (F.scaled_dot_product_attention)
 * reached from ldm_diffusers.py:297,436-440,528-534,553-559.
 * ------------------------------------------------------------------------------- */
typedef struct {
    int dtype;
    const void* q; const void* k; const void* v; void* o;
    int ldq, ldk, ldv, ldo;   /* row strides in elements */
    int B, H, Lq, Lk, D;
    float scale;
} madm_attention_args;
int madm_attention_fwd(const madm_attention_args* a, void* stream);

/* Backward of madm_attention_fwd (torch autograd through F.scaled_dot_product_attention in the reference,
 * engine/train_loop.py:203-217): dq / dk / dv (dtype, same addressing as q / k / v with their own row strides) from
 * q, k, v, the forward output o and its gradient dout.  Nothing is saved by the forward: the row log-sum-exp is
 * recomputed, together with rowsum(dout * o), into the f32 workspace (2 * B * H * Lq floats).  Two launches (dq, then
 * dk + dv), no atomics: deterministic.  D in {40, 64, 80, 160}. */
typedef struct {
    int dtype;
    const void* q; const void* k; const void* v; const void* o; const void* dout;
    void* dq; void* dk; void* dv;
    int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;   /* row strides in elements */
    int B, H, Lq, Lk, D;
    float scale;
    void* workspace;
    size_t workspace_bytes;
} madm_attention_bwd_args;
size_t madm_attention_bwd_workspace_bytes(const madm_attention_bwd_args* a);
int madm_attention_bwd(const madm_attention_bwd_args* a, void* stream);

/* p[r][:] = softmax(scale * s[r][:]) over L columns: f32 logits [rows][lds] -> dtype [rows][ldp].  With two
 * madm_conv2d_fwd GEMMs (S = Q K^T with out_f32, O = P V) this is the single-head d = 512, L = 4096
 * attention of the VAE mid block (ldm_diffusers.py:297), where GEMM tiles beat the flash kernel. */
int madm_softmax_rows(int dtype, const float* s, void* p, int rows, int L, int lds, int ldp,
                      float scale, void* stream);

/* ---------------------------------------------------------------------------------
 * Small glue kernels of LdmDiffusers.forward (ldm_diffusers.py:143-217).
 * ------------------------------------------------------------------------------- */
/* images [B,3,H,W] f32 in [0,1] (NCHW, as the backbone hands them over) ->
 * channels-last [B,H,W,Cpad] of dtype holding (x - mean)/std in channels 0..2, zeros above
 * (ldm_diffusers.py:145-146).  minmax[2] (f32, caller-initialised to +inf/-inf) receives the
 * normalised min / max so the host can honour the reference's range assert (:147) without
 * a mid-forward sync. */
int madm_image_to_nhwc(int dtype, const float* img, void* out, int B, int C, int H, int W,
                       int Cpad, float mean, float std, float* minmax, void* stream);

/* Same normalisation, but emitting the im2col rows of the VAE encoder's 3x3 / pad-1 stem conv
 * (vae.encoder.conv_in, ldm_diffusers.py:287): out[pixel][k], k = (kh*3 + kw)*3 + c for the 27 taps,
 * zero for k in [27, Kpad) and outside the image, so the 3->128 stem runs as a K = Kpad GEMM instead
 * of a 3x3 conv over channel-padded pixels (9x fewer MFMAs on a layer that is 95 % padding). */
int madm_image_to_im2col3x3(int dtype, const float* img, void* out, int B, int H, int W, int Kpad,
                            float mean, float std, float* minmax, void* stream);

/* The stem itself, without the detour: out[pixel][n] (dtype, row stride ldo) = bias[n] + sum_k wT[k][n] * x_k, x_k the
 * normalised image value of tap k = (kh*3 + kw)*3 + c (zero outside the image), N = 128 (the SD VAE's conv_in).  MADM_F32:
 * exact f32 FMAs; MADM_BF16 / MADM_F16 (ldo a multiple of 8): x_k and wT rounded to the dtype, f32 accumulation on the
 * matrix pipe (K = 27 padded to 32) -- what conv_in computes under the reference's autocast; wT f32 [27][N]; stats (NULL or f64 [B][N][2], zeroed by the caller) += per-(image, channel) sum / sum of squares
 * of the output (the statistics of the GroupNorm that follows, as the conv epilogues produce them); minmax as above.
 * Replaces vae.encoder.conv_in of vae_encoder (ldm_diffusers.py:287) together with the normalisation of :145-147. */
int madm_stem_conv3x3(int dtype, const float* img, const float* wT, const float* bias, void* out, int ldo, int B, int H,
                      int W, int N, float mean, float std, double* stats, float* minmax, void* stream);

/* moments [B*HW][ldm] (dtype; channels 0..3 = posterior mean) ->
 *   latents_nchw[B,4,h,w] f32 = mean * scaling_factor                     (ldm_diffusers.py:303-308)
 *   noisy[B*HW][Cpad] dtype   = sqrt_ac[t_b] * latents + sqrt_1mac[t_b] * noise   (:349-360)
 * noise is the shared [1,4,h,w] f32 tensor (seed 42, :73-75) broadcast over B; extra input
 * channels 4..Cpad-1 are zero (the K-tile padding of the UNet conv_in). */
int madm_latents_add_noise(int dtype, const void* moments, int ldm, float scaling,
                           const float* noise, const float* sqrt_ac, const float* sqrt_1mac,
                           const int64_t* timesteps, float* latents_nchw, void* noisy,
                           int B, int HW, int Cpad, void* stream);

/* diffusers Timesteps(320, flip_sin_to_cos=True, freq_shift=0): out[b] = [cos(t f_i) | sin(t f_i)],
 * f_i = exp(-ln(10000) i / half) handed over as the constant f32 table freqs[dim/2]; written as
 * dtype [B][dim]  (ldm_diffusers.py:498-503). */
int madm_timestep_embedding(int dtype, const int64_t* timesteps, const float* freqs, void* out,
                            int B, int dim, void* stream);

/* y = silu(x) elementwise over n elements of dtype (time_embedding.act, ResnetBlock2D
 * nonlinearity on temb). */
int madm_silu(int dtype, const void* x, void* y, size_t n, void* stream);

/* y[b][c] (f32) = x[b][c] (dtype) + add[b][c] (f32, may be NULL): "emb += res_time_embedding"
 * (ldm_diffusers.py:505-509) and the dtype->f32 hand-over of time rows to madm_conv2d_fwd. */
int madm_rows_to_f32(int dtype, const void* x, const float* add, float* y, size_t n, void* stream);

/* dst[r][0..C) = src[r][0..C) between row-strided tensors of dtype (widening the 4-channel UNet sample to
 * the K-tile-wide input of the VAE decoder's post_quant_conv, ldm_diffusers.py:319-320). */
int madm_copy_columns(int dtype, const void* src, int lds, void* dst, int ldd, size_t rows, int C,
                      void* stream);
/* y = clamp(x, lo, hi) on f32: torch.clip(decoder_output, -1, 1) of 'after_vae.decoder' (ldm_diffusers.py:214). */
int madm_clamp_f32(const float* x, float* y, size_t n, float lo, float hi, void* stream);

/* y (dtype) = x (f32), n elements: hand-over of f32 conditioning rows (cond_emb, prompt
 * embeddings; ldm_base.py:877-887) to the compute dtype. */
int madm_cast_from_f32(int dtype, const float* x, void* y, size_t n, void* stream);

/* channels-last [B*HW][ld] dtype (first C channels) -> channels [c_off, c_off + C) of the NCHW f32
 * tensor out[B, Ctot, H, W]: the tap / sample tensors returned to the detectron2-side consumers
 * (ldm_diffusers.py:209-217); c_off/Ctot let two sources form the concatenated 'in'-type taps. */
int madm_nhwc_to_nchw_f32(int dtype, const void* x, int ld, float* out, int B, int C, int c_off,
                          int Ctot, int HW, void* stream);
/* NCHW f32 [B,C,H,W] -> channels-last dtype [B*HW][Cpad] (zero padded). */
int madm_nchw_f32_to_nhwc(int dtype, const float* x, void* out, int B, int C, int HW, int Cpad,
                          void* stream);

/* ---------------------------------------------------------------------------------
 * Projection / segmentation-head rows.
 * ------------------------------------------------------------------------------- */
/* F.interpolate(mode='bilinear', align_corners=False) on channels-last tokens: in [B*IH*IW][ldi] (C channels)
 * -> out [B*OH*OW][ldo]; `out` may point into a wider concatenation buffer (DAFormerHead.forward resize + cat,
 * modeling/sem_seg_head/daformer_head.py:729-746; MTMADISE eval upsampling, mtmadise.py:687-691). */
int madm_resize_bilinear(int dtype, const void* in, int ldi, void* out, int ldo, int B, int IH, int IW,
                         int OH, int OW, int C, void* stream);
/* the same on `planes` NCHW f32 planes (backbone preprocess_image T.Resize, feature_extractor.py:77-79,140-146). */
int madm_resize_bilinear_nchw_f32(const float* in, float* out, int planes, int IH, int IW, int OH, int OW,
                                  void* stream);
/* out[pl][y][x] = in[pl][y1 + y][x1 + x] * scale inside the input, 0 outside, on `planes` f32 planes: "/255" + the
 * zero padding of ImageList.from_tensors (mtmadise.py:668-670), the final crop to the original size (:688) and the
 * sliding-window crops img[:, :, y1:y2, x1:x2] (feature_extractor.py:250). */
int madm_scale_pad_crop_nchw_f32(const float* in, float* out, int planes, int IH, int IW, int y1, int x1,
                                 int OH, int OW, float scale, void* stream);
/* sliding-window inference (feature_extractor.py:199-278): averages the nW (<= 4) windows' feature maps
 * win[nW][B][h][w][C] into the canvas out[B][h][Wc][C]; window k covers canvas columns [x1[k], x1[k] + w)
 * (x1 = host array); every canvas column is divided by the number of windows covering it. */
int madm_slide_merge(int dtype, const void* win, void* out, int nW, int B, int h, int w, int Wc, int C,
                     const int* x1, void* stream);
/* conf[(K+1) * pred[i] + gt'[i]] += 1, gt' = K where gt == ignore_label (int64 counters, caller-zeroed or running):
 * DSECSemSegEvaluator.process (evaluation/d2_evaluator.py:106-127); exact integer arithmetic. */
int madm_confusion_matrix(const int64_t* pred, const int64_t* gt, size_t n, int num_classes, int ignore_label,
                          int64_t* conf, void* stream);
/* depthwise 3x3 conv (dilation d, padding d, stride 1) on channels-last x [B*H*W][C] with w [9][C] f32
 * (tap-major), then y = act(acc * scale[c] + shift[c]) (scale/shift = eval-mode BatchNorm folded):
 * mmcv DepthwiseSeparableConvModule.depthwise_conv of the sep-ASPP (daformer_head.py:383-398). */
int madm_dwconv3x3(int dtype, const void* x, const float* w, const float* scale, const float* shift, void* y,
                   int ldy, int B, int H, int W, int C, int dilation, int act, void* stream);
/* out[r][i] = tanh(a1[i]) x1[i] + tanh(a2[i]) x2[i] for r < repeat (a1/a2 NULL = gate 1, x2 NULL = one term):
 * ClipFeatureProject.get_cond_prompt / get_cond_time + the batch repeat_interleave
 * (modeling/meta_arch/ldm_base.py:675-712,915-917). */
int madm_tanh_gate(const float* a1, const float* x1, const float* a2, const float* x2, float* out, size_t n,
                   int repeat, void* stream);
/* labels[b][p] = argmax_k logits[b][k][p] (first maximal k, torch.argmax semantics; bit-exact index op):
 * evaluation/d2_evaluator.py:106. */
int madm_argmax_nchw_f32(const float* x, int64_t* out, int B, int K, size_t HW, void* stream);

/* ---------------------------------------------------------------------------------
 * Backward of madm_conv2d_fwd.  The reference obtains these from torch autograd:
 * losses.backward() / scaler.scale(losses).backward() in engine/train_loop.py:203-217 through every nn.Conv2d /
 * nn.Linear / peft lora_A / lora_B executed under ldm_diffusers.py:283-616.
 *
 * Weight gradient:   dw[n][k] += sum_m dout[m][n] * A(m, k),  k = (kh, kw, c), A = the gather of madm_conv2d_fwd
 * (same geometry fields, incl. the two-source concat and the nearest-2x upsample).  The inputs are the tensors the
 * forward conv READ (a GroupNorm the forward fused must be materialised by the caller).  dw is f32 [N][K] dense and
 * is ACCUMULATED into (gradient accumulation; zero it for a plain gradient); the pixel range is split over `splitm`
 * grid slices (0 = choose) that add their partial tiles with float atomics, so the low-order bits depend on
 * scheduling, like torch's own non-deterministic wgrad algorithms.
 * bias gradient = column sums of dout: madm_colsum, or ``dbias`` below (the blocks of the first weight-column tile add up
 * the dout rows they stream anyway: no second pass over dout, no second launch).
 *
 * Data gradient of a stride-1 layer (3x3 / pad 1, 1x1, linear): din = madm_conv2d_fwd(dout, wt) with
 * pad' = KH - 1 - pad and wt[c][taps - 1 - t][n] = w[n][t][c], produced by madm_pack_dgrad_weights
 * (w dense [N][taps][C], wt dense [C][taps][N]).
 * ------------------------------------------------------------------------------- */
typedef struct {
    int dtype;            /* madm_dtype of in1 / in2 / dout */
    const void* in1;      /* forward input [B, IH, IW, C1] (pixel stride ld1) */
    const void* in2;      /* second source or NULL */
    int C1, C2;           /* multiples of 8 (bf16) / 4 (f32) elements */
    int ld1, ld2;         /* 0 = dense */
    const void* dout;     /* [M][ldd], M = B*OH*OW: gradient of the forward output */
    int ldd;              /* 0 = dense (N) */
    float* dw;            /* f32 [N][KH*KW*(C1+C2)], accumulated into */
    int B, IH, IW, OH, OW, KH, KW, stride, pad_t, pad_l, upsample;
    int N;                /* multiple of 4 */
    int splitm;           /* pixel slices; 0 = choose for the MI355X grid */
    float* dbias;         /* NULL or f32 [N], accumulated into: sum over the M rows of dout (nn.Conv2d / nn.Linear bias.grad) */
} madm_conv2d_wgrad_args;
int madm_conv2d_wgrad(const madm_conv2d_wgrad_args* a, void* stream);
int madm_pack_dgrad_weights(int dtype, const void* w, void* wt, int N, int taps, int C, void* stream);
/* f32 master weight (nn.Conv2d [N][Cin][KH][KW] with taps = KH * KW <= 9, nn.Linear: taps = 1) -> the packed forward
 * operand [N][taps][padded channels] of dtype, row stride ldo: the Cin channels are nsrc <= 4 concatenated sources of
 * src_channels[s] channels, each zero-padded to ktile (the K order of madm_conv2d_fwd's two-source gather).  interleave: out
 * row r <- w row (r & 1 ? N / 2 : 0) + r / 2 (diffusers GEGLU.proj: value_j / gate_j rows side by side).  The reference
 * keeps its weights in torch layouts (nn.Conv2d / nn.Linear inside diffusers, ldm_diffusers.py:60-75); after every
 * optimizer step (engine/train_loop.py:286) the packed copies are re-derived -- one launch per tensor. */
int madm_pack_weight(int dtype, const float* w, void* out, int ldo, int N, int Cin, int taps, int nsrc,
                     const int* src_channels, int ktile, int interleave, void* stream);
/* the operands of a LayerNorm-folded linear layer (madm_conv2d_args.ln_colsum) from the f32 masters, one launch:
 * out [N][K] = dtype(w * gamma), bias_out [N] = w beta + b (f64 accumulation; b may be NULL), colsum [N] = row sums of the
 * ROUNDED out (f64).  K = the normalised width, a multiple of the K tile; interleave as in madm_pack_weight. */
int madm_fold_layernorm_pack(int dtype, const float* w, const float* b, const float* gamma, const float* beta, void* out,
                             float* bias_out, float* colsum, int N, int K, int interleave, void* stream);
/* Data gradient of the other two conv geometries, reduced to the stride-1 case:
 *   stride-2 conv (UNet Downsample2D pad 1, VAE Downsample2D pad (0,1,0,1)): y [B][H][W][C] = dout [B][OH][OW][C]
 *     zero-inserted (y[2 oy][2 ox] = dout[oy][ox], 0 elsewhere), then the stride-1 data gradient with pad' = 2 - pad;
 *   nearest-2x upsample + conv (Upsample2D): the stride-1 data gradient at the upsampled size, then
 *     madm_sumpool2x2: y [B][H][W][C] = sum of the 2 x 2 blocks of x [B][2H][2W][C].
 * madm_silu_bwd: dx = dy * silu'(x) (time-embedding activation, ResnetBlock2D's silu(temb)). */
int madm_zero_insert2x(int dtype, const void* x, void* y, int B, int OH, int OW, int H, int W, int C, void* stream);
int madm_sumpool2x2(int dtype, const void* x, void* y, int B, int H, int W, int C, void* stream);
int madm_silu_bwd(int dtype, const void* x, const void* dy, void* dx, size_t n, void* stream);
/* y = a + b over n elements of dtype (n a multiple of 16 bytes): gradient accumulation where one tensor feeds two
 * consumers (UNet skip connections, the feature taps of ldm_diffusers.py:442-445). */
int madm_add(int dtype, const void* a, const void* b, void* y, size_t n, void* stream);
/* out[b][c] (f32 [B][C], accumulated into) += sum over the HW rows of image b of x[b*HW + r][c] (row stride ldx): the
 * bias gradient (B = 1 over all rows) and ResnetBlock2D's per-image time-row gradient. */
int madm_colsum(int dtype, const void* x, int ldx, int B, int HW, int C, float* out, void* stream);

/* Backward of madm_groupnorm_apply (without its residual input, whose gradient is dz itself) in two passes over
 * the source x [B*HW][C] occupying channels [c_off, c_off + C) of the Ctot-channel (possibly two-source) tensor; dy
 * [B*HW][lddy] is the gradient of the whole output (read at column c_off + c), sums1 / C1 / sums2 / gamma / beta /
 * eps / act exactly as in the forward call:
 *   bwd_sums : bsums (f64 [B][Ctot][2], zeroed by the caller) [b][c_off + c] += {sum_hw dz, sum_hw dz * xhat},
 *              dz = dy * act'(z); run it for every source before bwd_apply;
 *   bwd_apply: dx [B*HW][C] dense = rstd_g * (gamma * dz - mean_g(gamma dz) - xhat * mean_g(gamma dz xhat)), and, when
 *              dgamma / dbeta (f32 [Ctot], accumulated into) are given, dgamma[c] += sum_b S2, dbeta[c] += sum_b S1;
 *              dres (NULL or [B*HW][lddres], read at column c_off + c) is added to dx: the gradient that reaches x
 *              through a skip path (ResnetBlock2D's shortcut).
 * madm_layernorm_bwd: dx [M][C] of madm_layernorm_fwd (+ dres [M][C] or NULL: the gradient arriving through the
 * block's residual connection); dgamma / dbeta (f32 [C], accumulated into) may both be NULL.
 * torch autograd in the reference (engine/train_loop.py:203-217) through diffusers' GroupNorm / LayerNorm modules
 * (ldm_diffusers.py:290,297,299-300,387,435,553,609-610). */
int madm_groupnorm_bwd_sums(int dtype, const void* x, const void* dy, int lddy, int B, int HW, int C, int c_off, int Ctot,
                            int G, const double* sums1, int C1, const double* sums2, const float* gamma,
                            const float* beta, float eps, int act, double* bsums, void* stream);
int madm_groupnorm_bwd_apply(int dtype, const void* x, const void* dy, int lddy, void* dx, int B, int HW, int C, int c_off,
                             int Ctot, int G, const double* sums1, int C1, const double* sums2, const float* gamma,
                             const float* beta, float eps, int act, const double* bsums, float* dgamma, float* dbeta,
                             const void* dres, int lddres, void* stream);
int madm_layernorm_bwd(int dtype, const void* x, const void* dy, void* dx, int M, int C, const float* gamma, float eps,
                       float* dgamma, float* dbeta, const void* dres, void* stream);
/* GEGLU backward (diffusers GEGLU inside BasicTransformerBlock.ff): pre [M][N2] are the dense pre-activation rows of
 * the GEGLU GEMM in its interleaved (value_j, gate_j) order (madm_conv2d_fwd with MADM_EPI_NONE on the same packed
 * weights), dout [M][N2 / 2] the gradient of value * gelu_erf(gate); dpre [M][N2] in the same interleaved order. */
int madm_geglu_bwd(int dtype, const void* pre, const void* dout, void* dpre, size_t M, int N2, void* stream);

/* ---------------------------------------------------------------------------------
 * Training-step tail on ONE flat, 16-byte-aligned fp32 buffer per role (HBM-bound).
 * ------------------------------------------------------------------------------- */
/* *out (f64, caller-zeroed) += sum x[i]^2 : the total norm of torch.nn.utils.clip_grad_norm_
 * (engine/train_loop.py:203-217). */
int madm_sumsq_f32(const float* x, size_t n, double* out, void* stream);
/* torch.optim.AdamW step number `step` (>= 1) on p/g/m/v[n] (config_files/common/optim.py:8-17): the gradient is
 * first multiplied by grad_scale = (1 / loss_scale) * min(1, clip / (norm + 1e-6)) -- GradScaler.unscale_ and
 * clip_grad_norm_ folded into the same pass (28 bytes of HBM traffic per element instead of three passes). */
int madm_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, float grad_scale, void* stream);
/* ema = alpha * ema + (1 - alpha) * p : CMDISE._update_ema (modeling/meta_arch/cmdise.py:337-349), one launch for all
 * teacher parameters instead of a Python loop over tensors. */
/* The same step with PER-TENSOR hyper-parameters (get_default_optimizer_params_unet, utils/parameter_count.py:120-215:
 * weight_decay_norm = weight_decay_bias = 0, optional unet_lr; torch.optim.AdamW skipping parameters without a gradient):
 * the flat buffers are laid out in 1024-element chunks that never straddle two tensors, chunk_tensor[n / 1024] (int32) names
 * each chunk's tensor and hyper[tensor] = {lr, weight_decay, 1 - beta1^step_t, sqrt(1 - beta2^step_t)} (f32 x 4, 16-byte
 * aligned; a zero third entry skips the tensor this step). */
int madm_adamw_step_table(float* p, const float* g, float* m, float* v, size_t n, const int* chunk_tensor, const float* hyper,
                          float beta1, float beta2, float eps, float grad_scale, void* stream);
int madm_ema_update(float* ema, const float* p, size_t n, float alpha, void* stream);

/* ---- label / pseudo-label pipeline of the self-training step, on the device (bit-exact index work) --------------
 * MTMADISE.convert_label_to_rgb (modeling/meta_arch/mtmadise.py:159-175: label.cpu() -> PIL 'P' image ->
 * putpalette -> RGB -> (x / 255 - 0.5) / 0.5, valid = label != 255): label i64 [B][HW] (values taken mod 256 like
 * numpy's astype(uint8)), palette768 = the zero-padded 256 x 3 byte palette (:97-103), rgb f32 [B][3][HW],
 * valid f32 [B][HW] or NULL. */
int madm_label_to_rgb(const int64_t* label, const unsigned char* palette768, float* rgb, float* valid, int B, int HW,
                      void* stream);
/* pseudo labels (mtmadise.py:340-348): prob [B][HW] = max_k softmax(logits [B][K][HW]), label = first argmax (i64),
 * *count_ge (u64, zeroed by the caller, NULL = skip) += number of pixels with prob >= threshold -- the reference's
 * torch.sum(...).item() / size without the host sync. */
int madm_pseudo_label(const float* logits, float* prob, int64_t* label, unsigned long long* count_ge, int B, int K, int HW,
                      float threshold, void* stream);
/* presence256[v] = 1 for every value v = label & 255 that occurs (the class list of torch.unique in
 * utils/dacs_transforms.py:84 without the sort); presence256 zeroed by the caller. */
int madm_label_presence(const int64_t* label, size_t n, unsigned* presence256, void* stream);
/* ClassMix of ONE image pair (utils/dacs_transforms.py:92-111): mask = label0 in {v : chosen256[v]};
 * out = mask * x0 + (1 - mask) * x1 for the C f32 planes [C][HW] and the i64 label plane; any of mask_out / img_out /
 * label_out may be NULL. */
int madm_class_mix(const int64_t* label0, const int64_t* label1, const unsigned char* chosen256, const float* img0,
                   const float* img1, int C, int HW, float* mask_out, float* img_out, int64_t* label_out, void* stream);

/* ---------------------------------------------------------------------------------
 * Training step of the meta-architecture (BASELINE config 4): what torch autograd does
 * behind ``losses.backward()`` (engine/train_loop.py:203-217) for the train-mode DAFormer head
 * (modeling/sem_seg_head/daformer_head.py:677-749), the CmdiseCriterion losses (modeling/criterion.py:120-131,155-254) and
 * the prompt / time gates (modeling/meta_arch/ldm_base.py:675-712).  Train-mode BatchNorm needs no entry point of its
 * own: over channels-last tokens it IS madm_groupnorm_* with B = 1, HW = all rows of the batch, G = C (per-channel
 * statistics over batch and pixels, biased variance) -- forward, fused activation and backward included.
 * ------------------------------------------------------------------------------- */
/* y[b][p][c] = x[b][p][c] * scale[b][c] (f32 [B][C]): nn.Dropout2d(0.1) with scale = keep / (1 - p) per (image, channel)
 * (daformer_head.py:677-699 cls_seg); the same call on the gradient is its backward.  x / y row strides ldx / ldy. */
int madm_scale_channels(int dtype, const void* x, int ldx, const float* scale, void* y, int ldy, int B, int HW, int C,
                        void* stream);
/* train-mode nn.BatchNorm2d bookkeeping (mmcv ConvModule norm, daformer_head.py:364-372,455-461): folds the per-image
 * channel sums chsums [B][C][2] (a conv epilogue's / madm_groupnorm_stats') into the batch sums st [1][C][2] that
 * madm_groupnorm_* consume with B = 1, G = C, and updates running_mean / running_var (f32 [C], NULL = skip) with
 * momentum and the unbiased variance over `count` = B * H * W elements. */
int madm_bn_fold_stats(const double* chsums, int B, int C, double count, float momentum, double* st, float* running_mean,
                       float* running_var, void* stream);
/* dx = dy where y > 0 else 0, y = the ReLU's output (d2 BottleneckBlock's relu(GN(conv3) + shortcut),
 * modeling/backbone/feature_extractor.py:347-359); n elements, dense. */
int madm_relu_bwd(int dtype, const void* y, const void* dy, void* dx, size_t n, void* stream);
/* weight gradient of madm_dwconv3x3 (before its affine / activation): dw (f32 [9][C], accumulated into)
 * [t][c] += sum_pixels dy[p][c] * x[p + offset_t][c]; x dense [B*H*W][C], dy row stride lddy.  The DATA gradient of the
 * depthwise conv is madm_dwconv3x3 itself on dy with the taps reversed (symmetric dilation). */
int madm_dwconv3x3_wgrad(int dtype, const void* x, const void* dy, int lddy, float* dw, int B, int H, int W, int C,
                         int dilation, void* stream);
/* adjoint of madm_resize_bilinear: dout [B*OH*OW][lddo] (C channels) -> din [B*IH*IW][C] dense, as two 1-D passes with
 * an f32 intermediate [B*OH*IW][C] in `workspace` (gather form: deterministic, no atomics). */
size_t madm_resize_bilinear_bwd_workspace_bytes(int B, int IW, int OH, int C);
int madm_resize_bilinear_bwd(int dtype, const void* dout, int lddo, void* din, int B, int IH, int IW, int OH, int OW, int C,
                             void* workspace, size_t workspace_bytes, void* stream);
/* CmdiseCriterion.cross_entropy (criterion.py:120-131): F.cross_entropy(reduction='none', ignore_index) * pixel_weight on
 * f32 logit tokens [M][ldx] (K classes), labels i64 [M], weight f32 [M] or NULL:
 *   *loss_sum (f64, caller-zeroed, NULL = skip) += sum_i w_i (logsumexp(x_i) - x_i[label_i]) over the non-ignored pixels;
 *   dlogits (NULL = skip; `dtype` tokens [M][ldd], columns >= K zero) = coef * (*gscale, NULL = 1) * w_i *
 *   (softmax(x_i) - onehot(label_i)); the caller folds the mean's 1 / M and the loss weight into coef, gscale is the
 *   upstream gradient of the loss scalar (GradScaler scale included) read on the device. */
int madm_softmax_ce(int dtype, const float* logits, int ldx, int K, const int64_t* labels, const float* weight,
                    int ignore_index, size_t M, double* loss_sum, const float* gscale, float coef, void* dlogits, int ldd,
                    void* stream);
/* vae_decoder / mic / denoise losses (criterion.py:222-254): pred, gt NCHW f32 [B][C][h][w]; mask f32 [B][Hm][Wm] read
 * through F.interpolate(mode='nearest') to (h, w) and broadcast over C (NULL = 1):
 *   *loss_sum (f64) += sum |pred - gt| * mask (l2 = 0) or sum (pred - gt)^2 * mask (l2 = 1);
 *   dpred (NULL = skip) = coef * (*gscale) * mask * sign(pred - gt)  (or 2 (pred - gt)). */
int madm_masked_l1(const float* pred, const float* gt, const float* mask, int B, int C, int h, int w, int Hm, int Wm, int l2,
                   double* loss_sum, const float* gscale, float coef, float* dpred, void* stream);
/* strong_transform's colour augmentation of the mixed / target image (utils/dacs_transforms.py:40-78; kornia
 * ColorJitter + GaussianBlur2d, un-vendored and unpinned -- restated in oracle/augment.py) on ONE RGB image [3][HW] f32 in
 * [0, 1]: madm_gray_sum adds sum(0.299 r + 0.587 g + 0.114 b) into *out (f64); madm_color_jitter_step applies one of
 * 0 brightness (x f), 1 contrast (x f + mean_gray (1 - f), mean from *gray_sum / HW), 2 saturation ((1 - f) gray + f x),
 * 3 hue (HSV rotation by f radians); madm_blur_axis_f32 is one pass of the separable Gaussian blur (reflect border) along
 * axis 0 (y) or 1 (x) with `ksize` device weights. */
int madm_gray_sum(const float* img, size_t HW, double* out, void* stream);
int madm_color_jitter_step(const float* in, float* out, size_t HW, int op, float factor, const double* gray_sum, void* stream);
int madm_blur_axis_f32(const float* in, float* out, int planes, int H, int W, int axis, int ksize, const float* weights,
                       void* stream);
/* backward of madm_tanh_gate for dout [repeat][n]: with g_i = sum_r dout[r][i], dx1 += tanh(a1) g, da1 += (1 - tanh^2(a1))
 * x1 g and likewise for (a2, x2); every output is ACCUMULATED into and may be NULL. */
int madm_tanh_gate_bwd(const float* a1, const float* x1, const float* a2, const float* x2, const float* dout, float* da1,
                       float* dx1, float* da2, float* dx2, size_t n, int repeat, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MADM_HIP_H */
