// Glue kernels of LdmDiffusers.forward: layout changes, image normalisation, latent scaling +
// noise mixing, sinusoidal timestep embedding, SiLU on time rows.  All HBM/launch-bound.
#include "common.hpp"

namespace {

// one atomicMin/atomicMax pair per BLOCK (thousands of waves hammering two addresses serialise)
__device__ __forceinline__ void block_minmax(float lo, float hi, float* minmax) {
    __shared__ float s_lo[4], s_hi[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_lo[w] = lo; s_hi[w] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 1; i < nw; ++i) { lo = fminf(lo, s_lo[i]); hi = fmaxf(hi, s_hi[i]); }
        // Native integer atomics on the float bit patterns (sign-aware: for v >= 0 the signed-int order is the float
        // order, for v < 0 the unsigned order is the reversed float order).  atomicMin/Max(float*) compile to CAS
        // loops: with the first ~2000 resident blocks all improving the freshly reset bound they serialised into
        // ~0.5 ms per call.  A block that cannot improve the published bound skips its atomic altogether.
        if (lo <= hi) {
            if (lo < __builtin_nontemporal_load(&minmax[0])) {
                if (lo >= 0.f) atomicMin(reinterpret_cast<int*>(&minmax[0]), __float_as_int(lo));
                else atomicMax(reinterpret_cast<unsigned*>(&minmax[0]), __float_as_uint(lo));
            }
            if (hi > __builtin_nontemporal_load(&minmax[1])) {
                if (hi >= 0.f) atomicMax(reinterpret_cast<int*>(&minmax[1]), __float_as_int(hi));
                else atomicMin(reinterpret_cast<unsigned*>(&minmax[1]), __float_as_uint(hi));
            }
        }
    }
}

// min / max of (x - mean) * inv_std over a flat f32 buffer, folded into minmax[2] (the reference's input-range
// assert, ldm_diffusers.py:147).  A FEW blocks only: agent-scope atomics on one address execute at the memory side
// at ~85 ns each, so one atomic pair per block of the image kernels (8192 blocks) cost 0.5-0.7 ms per call --
// far more than the transform itself; 64 blocks re-read the (L2/MALL-resident) image instead.
__global__ __launch_bounds__(256) void range_probe_kernel(const float* __restrict__ x, size_t n, float mean,
                                                          float inv_std, float* minmax) {
    float lo = INFINITY, hi = -INFINITY;
    const size_t n4 = n / 4;
    // min / max commute with the (monotone) normalisation: taken on the raw values, normalised once at the end; eight loads
    // in flight per thread (one load per trip: 24 dependent ~0.5 us round trips = the 12 us this kernel took for a 6 MB batch)
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(x)[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            lo = fminf(fminf(lo, v[u].x), fminf(fminf(v[u].y, v[u].z), v[u].w));
            hi = fmaxf(fmaxf(hi, v[u].x), fmaxf(fmaxf(v[u].y, v[u].z), v[u].w));
        }
    }
    for (; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        lo = fminf(fminf(lo, v.x), fminf(fminf(v.y, v.z), v.w));
        hi = fmaxf(fmaxf(hi, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
    }
    if (lo <= hi) {   // (a thread that saw no element keeps its +inf / -inf); a negative std swaps the roles
        const float a = (lo - mean) * inv_std, b = (hi - mean) * inv_std;
        lo = fminf(a, b); hi = fmaxf(a, b);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float a = (x[n4 * 4 + threadIdx.x] - mean) * inv_std;
        lo = fminf(lo, a);
        hi = fmaxf(hi, a);
    }
    block_minmax(lo, hi, minmax);
}

// NCHW f32 image -> channels-last dtype, (x - mean) / std in the first C channels, zeros above.
template <typename T>
__global__ __launch_bounds__(256) void image_to_nhwc_kernel(const float* __restrict__ img, T* __restrict__ out,
                                                            int B, int C, int HW, int Cpad, float mean,
                                                            float inv_std) {
    constexpr int EPC = TT<T>::EPC;
    const unsigned total = (unsigned)B * (unsigned)HW;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int b = (int)(idx / (unsigned)HW);
        const int px = (int)(idx - (unsigned)b * (unsigned)HW);
        T* o = out + (size_t)idx * Cpad;
        for (int q = 0; q < Cpad / EPC; ++q) {
            float f[EPC];
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                const int c = q * EPC + j;
                float v = 0.f;
                if (c < C) v = (img[((size_t)b * C + c) * HW + px] - mean) * inv_std;
                f[j] = v;
            }
            *reinterpret_cast<uint4*>(o + q * EPC) = f32_to_chunk<T>(f);
        }
    }
}

// NCHW f32 3-channel image -> im2col rows of the 3x3/pad-1 stem conv: out[pixel][k], k = (r*3+s)*3 + c
// holding (img[c, y+r-1, x+s-1] - mean) / std (zero outside the image: the conv pads the NORMALISED
// image), k >= 27 zero.  One block per 64-pixel row segment: the 3 planes x 3 rows x 66 normalised inputs go
// through LDS (coalesced row reads), then every thread assembles 16-byte chunks and consecutive threads write
// consecutive 16 bytes -- the kernel is bound by the Kpad * sizeof(T) bytes it writes per pixel.
constexpr int IM2COL_PX = 64;
template <typename T>
__global__ __launch_bounds__(256) void image_to_im2col_kernel(const float* __restrict__ img, T* __restrict__ out,
                                                              int B, int H, int W, int Kpad, float mean,
                                                              float inv_std) {
    constexpr int EPC = TT<T>::EPC;
    constexpr int TW = IM2COL_PX + 2;
    __shared__ float tile[9 * TW];   // [c][r][x]
    const unsigned segs = (unsigned)(W + IM2COL_PX - 1) / IM2COL_PX;
    const unsigned seg = blockIdx.x % segs, row = blockIdx.x / segs;   // row = b * H + y
    const int b = (int)(row / (unsigned)H), y = (int)(row - (unsigned)b * H), x0 = (int)seg * IM2COL_PX;
    const size_t HW = (size_t)H * W;
    const float* ib = img + (size_t)b * 3 * HW;
    for (int i = threadIdx.x; i < 9 * TW; i += 256) {
        const int cr = i / TW, xx = i - cr * TW;
        const int c = cr / 3, r = cr - 3 * c;
        const int yy = y + r - 1, gx = x0 + xx - 1;
        float t = 0.f;
        if ((unsigned)yy < (unsigned)H && (unsigned)gx < (unsigned)W)
            t = (ib[(size_t)c * HW + (size_t)yy * W + gx] - mean) * inv_std;
        tile[i] = t;
    }
    __syncthreads();
    const int CPR = Kpad / EPC;
    for (int i = threadIdx.x; i < IM2COL_PX * CPR; i += 256) {
        const int p = i / CPR, q = i - p * CPR;
        if (x0 + p >= W) break;
        float v[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            const int k = q * EPC + j;          // k = (r * 3 + s) * 3 + c
            const int tap = k / 3, c = k - 3 * tap;
            const int r = tap / 3, s2 = tap - 3 * r;
            v[j] = (k < 27) ? tile[(c * 3 + r) * TW + p + s2] : 0.f;
        }
        *reinterpret_cast<uint4*>(out + ((size_t)row * W + x0 + p) * Kpad + q * EPC) = f32_to_chunk<T>(v);
    }
}

// ---- the VAE encoder's stem, vae.encoder.conv_in (3 -> 128 channels, 3x3 / pad 1, ldm_diffusers.py:287), straight from
// the f32 NCHW image: normalise, convolve, add the bias, round to T, and sum the output's GroupNorm statistics.  27 inputs
// per pixel do not feed an MFMA tile; as im2col rows + GEMM the layer wrote and re-read 67 MB of rows for 6 MB of image
// (31 + 91 us).  Here: block = 4 image rows x 128 pixels; the 3 x 6 x 130 normalised input values sit in LDS; a group
// of 32 lanes owns a pixel pair, lane l its channels 4 l .. 4 l + 3 with the 27 x 4 weights in registers (exact f32 FMAs,
// packed), so the 32 lanes write full 256-byte lines (f32: 512 B).  HBM: 6 MB in, 134 MB out.  Several rows per block
// because of the fused statistics: one f64 atomic per (block, channel, moment) lands on 512 addresses, and 2 048 atomics
// on one address (one row per block) took as long as the two kernels this one replaces (122 us; they execute at the
// memory side at ~50-85 ns each).  Measured in a graph loop, 2 x 512 x 512: 93.5 / 88.7 us with 2 / 4 rows per block.
constexpr int STEM_SEG = 128, STEM_N = 128, STEM_ROWS = 4, STEM_MFMA_ROWS = 8;
template <typename T>
__global__ __launch_bounds__(256) void stem_conv3x3_kernel(const float* __restrict__ img, const float* __restrict__ wT,
                                                           const float* __restrict__ bias, T* __restrict__ out, int ldo,
                                                           int B, int H, int W, float mean, float inv_std,
                                                           double* __restrict__ stats) {
    constexpr int TW = STEM_SEG + 2, TR = STEM_ROWS + 2;
    __shared__ float tile[3 * TR * TW];            // [c][input row][x], normalised, zero outside the image
    __shared__ float red[8 * STEM_N * 2];
    const int tid = threadIdx.x, slot = tid >> 5, l = tid & 31;
    const unsigned segs = (unsigned)(W + STEM_SEG - 1) / STEM_SEG;
    const unsigned bands = (unsigned)(H + STEM_ROWS - 1) / STEM_ROWS;
    const unsigned seg = blockIdx.x % segs, bb = blockIdx.x / segs;
    const int b = (int)(bb / bands), y0 = (int)(bb - (unsigned)b * bands) * STEM_ROWS, x0 = (int)seg * STEM_SEG;
    const size_t HW = (size_t)H * W;
    const float* ib = img + (size_t)b * 3 * HW;
    for (int i = tid; i < 3 * TR * TW; i += 256) {
        const int cr = i / TW, xx = i - cr * TW;
        const int c = cr / TR, r = cr - TR * c;
        const int yy = y0 + r - 1, gx = x0 + xx - 1;
        float t = 0.f;
        if ((unsigned)yy < (unsigned)H && (unsigned)gx < (unsigned)W)
            t = (ib[(size_t)c * HW + (size_t)yy * W + gx] - mean) * inv_std;
        tile[i] = t;
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 wlo[27], whi[27];                        // wT [27][128], k = (r * 3 + s) * 3 + c: channels 4 l, 4 l + 1 | + 2, + 3
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(wT + k * STEM_N + 4 * l);
        wlo[k] = f32x2{t.x, t.y};
        whi[k] = f32x2{t.z, t.w};
    }
    const float4 bv = *reinterpret_cast<const float4*>(bias + 4 * l);
    __syncthreads();
    f32x2 smlo = {0.f, 0.f}, smhi = {0.f, 0.f}, sqlo = {0.f, 0.f}, sqhi = {0.f, 0.f};
    // two neighbouring pixels per step: their 3 x 3 windows share 2 of 3 columns (one 4-float read per (channel, row)
    // instead of 2 x 3), and every weight pair feeds two packed FMAs
    for (int pp = slot; pp < STEM_ROWS * (STEM_SEG / 2); pp += 8) {
        const int ry = pp / (STEM_SEG / 2), p = 2 * (pp - ry * (STEM_SEG / 2));
        const int y = y0 + ry;
        if (y >= H) break;
        if (x0 + p >= W) continue;
        f32x2 a0lo = {bv.x, bv.y}, a0hi = {bv.z, bv.w}, a1lo = a0lo, a1hi = a0hi;
#pragma unroll
        for (int cr = 0; cr < 9; ++cr) {           // cr = c * 3 + r
            const int c = cr / 3, r = cr - 3 * c;
            const float* tp = tile + (c * TR + ry + r) * TW + p;                            // TW and p are even
            const float2 x01 = *reinterpret_cast<const float2*>(tp);
            const float2 x23 = *reinterpret_cast<const float2*>(tp + 2);
            const float xs[4] = {x01.x, x01.y, x23.x, x23.y};
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                const int k = (r * 3 + s2) * 3 + c;
                const f32x2 v0 = {xs[s2], xs[s2]}, v1 = {xs[s2 + 1], xs[s2 + 1]};
                a0lo = __builtin_elementwise_fma(v0, wlo[k], a0lo);
                a0hi = __builtin_elementwise_fma(v0, whi[k], a0hi);
                a1lo = __builtin_elementwise_fma(v1, wlo[k], a1lo);
                a1hi = __builtin_elementwise_fma(v1, whi[k], a1hi);
            }
        }
        T* o = out + (((size_t)b * H + y) * W + x0 + p) * ldo + 4 * l;
        store4<T>(o, f32x4{a0lo[0], a0lo[1], a0hi[0], a0hi[1]});
        smlo += a0lo; smhi += a0hi;
        sqlo = __builtin_elementwise_fma(a0lo, a0lo, sqlo); sqhi = __builtin_elementwise_fma(a0hi, a0hi, sqhi);
        if (x0 + p + 1 < W) {
            store4<T>(o + ldo, f32x4{a1lo[0], a1lo[1], a1hi[0], a1hi[1]});
            smlo += a1lo; smhi += a1hi;
            sqlo = __builtin_elementwise_fma(a1lo, a1lo, sqlo); sqhi = __builtin_elementwise_fma(a1hi, a1hi, sqhi);
        }
    }
    const float4 sm = make_float4(smlo[0], smlo[1], smhi[0], smhi[1]), sq = make_float4(sqlo[0], sqlo[1], sqhi[0], sqhi[1]);
    if (stats) {
        float* rs = red + (slot * STEM_N + 4 * l) * 2;
        rs[0] = sm.x; rs[1] = sq.x; rs[2] = sm.y; rs[3] = sq.y; rs[4] = sm.z; rs[5] = sq.z; rs[6] = sm.w; rs[7] = sq.w;
        __syncthreads();
        float t = 0.f;                             // tid = channel * 2 + {sum, sum of squares}
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g * STEM_N * 2 + tid];
        atomicAdd(stats + (size_t)b * STEM_N * 2 + tid, (double)t);
    }
}

// sum over the 16 lanes of a DPP row, result in every lane (igemm_common.hpp row16_sum)
__device__ __forceinline__ float row16_sum_f(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));
    return x;
}

// The same stem on the matrix pipe for the 16-bit types: the reference runs conv_in under fp16 autocast (image and weights
// rounded to fp16, fp32 accumulation -- exactly an MFMA with K = 27 padded to 32), so nothing is lost against the exact-f32
// FMAs above, which were VALU-bound (216 FMAs per pixel pair and lane: 78 us for 134 MB of output).  Block = 4 rows x 128
// pixels, one row per wave = 8 tiles of 16 pixels; per tile the lane (pixel lane & 15, k group lane >> 4) gathers its 8
// im2col values (k = (r * 3 + s) * 3 + c) from the normalised 16-bit window in LDS, 8 MFMAs give its pixel 8 x 4 channels,
// which go through a wave-private padded LDS tile into full 256-byte lines.  Output statistics as above (f32 values
// before rounding, f32 block partials, f64 atomics).
template <typename T>
__global__ __launch_bounds__(256) void stem_conv3x3_mfma_kernel(const float* __restrict__ img, const float* __restrict__ wT,
                                                                const float* __restrict__ bias, T* __restrict__ out, int ldo,
                                                                int B, int H, int W, float mean, float inv_std,
                                                                double* __restrict__ stats) {
    static_assert(sizeof(T) == 2, "stem_conv3x3_mfma_kernel: 16-bit types");
    constexpr int MROWS = STEM_MFMA_ROWS;                         // rows per block: two per wave
    constexpr int TW = STEM_SEG + 2, TR = MROWS + 2;
    constexpr int ROWB = STEM_N * 2 + 16;                        // padded row of the output staging tile
    __shared__ __attribute__((aligned(16))) T tile[3 * TR * TW + 8];   // [c][input row][x], normalised, zero outside; + zeros for k >= 27
    __shared__ __attribute__((aligned(16))) char stage[4 * 16 * ROWB]; // per wave: 16 pixels x 128 channels
    __shared__ float red[4 * STEM_N * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const unsigned segs = (unsigned)(W + STEM_SEG - 1) / STEM_SEG;
    const unsigned bands = (unsigned)(H + MROWS - 1) / MROWS;
    const unsigned seg = blockIdx.x % segs, bb = blockIdx.x / segs;
    const int b = (int)(bb / bands), y0 = (int)(bb - (unsigned)b * bands) * MROWS, x0 = (int)seg * STEM_SEG;
    const size_t HW = (size_t)H * W;
    const float* ib = img + (size_t)b * 3 * HW;
    // window: row-major walk without divisions (18 rows of 130: thread t takes columns t, t + 256 of a row pair ...)
    for (int cr = wave; cr < 3 * TR; cr += 4) {
        const int c = cr / TR, r = cr - TR * c;
        const int yy = y0 + r - 1;
        for (int xx = lane; xx < TW; xx += 64) {
            const int gx = x0 + xx - 1;
            float t = 0.f;
            if ((unsigned)yy < (unsigned)H && (unsigned)gx < (unsigned)W)
                t = (ib[(size_t)c * HW + (size_t)yy * W + gx] - mean) * inv_std;
            TT<T>::st(&tile[cr * TW + xx], t);
        }
    }
    if (tid < 8) TT<T>::st(&tile[3 * TR * TW + tid], 0.f);
    // weight fragments: lane = (channel frow of the 16-channel tile, k group fg): w[k = 8 fg + j][n = 16 t + frow]; the
    // 27 x 128 f32 table comes in by coalesced float4 loads through the (not yet used) staging tile
    {
        float* wl = reinterpret_cast<float*>(stage);
        for (int i = tid; i < 27 * STEM_N / 4; i += 256)
            *reinterpret_cast<float4*>(wl + 4 * i) = *reinterpret_cast<const float4*>(wT + 4 * i);
    }
    __syncthreads();
    uint4 wf[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float* wl = reinterpret_cast<const float*>(stage);
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = fg * 8 + j;
            f[j] = k < 27 ? wl[k * STEM_N + t * 16 + frow] : 0.f;
        }
        wf[t] = f32_to_chunk<T>(f);
    }
    f32x4 bv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float4 q = *reinterpret_cast<const float4*>(bias + t * 16 + fg * 4);
        bv[t] = f32x4{q.x, q.y, q.z, q.w};
    }
    // the lane's 8 gather offsets (elements, relative to the window element of its pixel): k -> (tap r, s; channel c)
    int goff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = fg * 8 + j;
        const int tap = k / 3, c = k - 3 * tap, r = tap / 3, sx = tap - 3 * r;
        goff[j] = k < 27 ? (c * TR + r) * TW + sx : -1;
    }
    __syncthreads();
    f32x4 cs[8], cq[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { cs[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cq[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    char* st = stage + wave * 16 * ROWB;
    const unsigned short* tw16 = reinterpret_cast<const unsigned short*>(tile);
    const int nxt = (W - x0 + 15) / 16 < STEM_SEG / 16 ? (W - x0 + 15) / 16 : STEM_SEG / 16;   // wave-uniform
#pragma unroll 1
    for (int ry = wave; ry < MROWS; ry += 4) {
        const int y = y0 + ry;
        if (y >= H) break;
#pragma unroll 1
        for (int xt = 0; xt < nxt; ++xt) {
            const int px = xt * 16 + frow;                 // pixel of this lane inside the segment
            // A fragment: 8 window elements of pixel px (window row ry + r, column px + s)
            const int pbase = ry * TW + px;
            unsigned e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = tw16[goff[j] >= 0 ? pbase + goff[j] : 3 * TR * TW];
            const uint4 af = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
            const bool inside = x0 + px < W;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                f32x4 acc = bv[t];
                mma16<T>(wf[t], af, acc);                 // lane: pixel frow, channels 16 t + 4 fg .. + 3
                if (inside) { cs[t] += acc; cq[t] += acc * acc; }
                store4<T>(reinterpret_cast<T*>(st + frow * ROWB) + t * 16 + fg * 4, acc);
            }
            // the wave's 16 x 256-byte tile -> four instructions of 4 pixel rows x 16 lanes x 16 bytes (full lines)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int p = it * 4 + (lane >> 4), ch = lane & 15;
                const uint4 v = *reinterpret_cast<const uint4*>(st + p * ROWB + ch * 16);
                if (x0 + xt * 16 + p < W)
                    *reinterpret_cast<uint4*>(out + (((size_t)b * H + y) * W + x0 + xt * 16 + p) * ldo + ch * 8) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (stats) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = row16_sum_f(cs[t][r]), q = row16_sum_f(cq[t][r]);
                if (frow == 0) {
                    red[(wave * STEM_N + t * 16 + fg * 4 + r) * 2] = a;
                    red[(wave * STEM_N + t * 16 + fg * 4 + r) * 2 + 1] = q;
                }
            }
        __syncthreads();
        float t = 0.f;                                     // tid = channel * 2 + {sum, sum of squares}
#pragma unroll
        for (int g = 0; g < 4; ++g) t += red[g * STEM_N * 2 + tid];
        atomicAdd(stats + (size_t)b * STEM_N * 2 + tid, (double)t);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void latents_add_noise_kernel(const T* __restrict__ moments, int ldm,
                                                                float scaling, const float* __restrict__ noise,
                                                                const float* __restrict__ sqrt_ac,
                                                                const float* __restrict__ sqrt_1mac,
                                                                const int64_t* __restrict__ timesteps,
                                                                float* __restrict__ latents_nchw,
                                                                T* __restrict__ noisy, int B, int HW, int Cpad) {
    constexpr int EPC = TT<T>::EPC;
    const size_t total = (size_t)B * HW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / HW);
        const int px = (int)(idx - (size_t)b * HW);
        const int t = (int)timesteps[b];
        const float sa = sqrt_ac[t], sn = sqrt_1mac[t];
        float lat[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            lat[c] = TT<T>::ld(moments + idx * ldm + c) * scaling;
            if (latents_nchw) latents_nchw[((size_t)b * 4 + c) * HW + px] = lat[c];
        }
        T* o = noisy + idx * Cpad;
        for (int q = 0; q < Cpad / EPC; ++q) {
            float f[EPC];
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                const int c = q * EPC + j;
                f[j] = (c < 4) ? sa * lat[c] + sn * noise[(size_t)c * HW + px] : 0.f;
            }
            *reinterpret_cast<uint4*>(o + q * EPC) = f32_to_chunk<T>(f);
        }
    }
}

template <typename T>
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ timesteps, const float* __restrict__ freqs,
                                          T* __restrict__ out, int B, int dim) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * half) return;
    const int b = idx / half, i = idx - b * half;
    const float arg = (float)timesteps[b] * freqs[i];
    TT<T>::st(out + (size_t)b * dim + i, cosf(arg));
    TT<T>::st(out + (size_t)b * dim + half + i, sinf(arg));
}

template <typename T>
__global__ void silu_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        TT<T>::st(y + i, silu_f(TT<T>::ld(x + i)));
}

template <typename T>
__global__ void rows_to_f32_kernel(const T* __restrict__ x, const float* __restrict__ add, float* __restrict__ y,
                                   size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = TT<T>::ld(x + i) + (add ? add[i] : 0.f);
}

template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ x, T* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        TT<T>::st(y + i, x[i]);
}

// dst[r][0..C) = src[r][0..C) for two row-strided 2-D tensors of the same dtype (tiny channel counts:
// widening the 4-channel latent to a K-tile-wide buffer)
template <typename T>
__global__ void copy_columns_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, size_t rows,
                                    int C) {
    const size_t total = rows * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C;
        const int c = (int)(i - r * C);
        dst[r * ldd + c] = src[r * lds_ + c];
    }
}

__global__ void clamp_f32_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float lo, float hi) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = fminf(fmaxf(x[i], lo), hi);
}

// [B*HW][ld] (first C channels) -> channels [c_off, c_off + C) of [B][Ctot][HW] f32, 32x32 LDS tile transpose
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ x, int ld, float* __restrict__ out,
                                                           int C, int c_off, int Ctot, int HW) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int hw = hw0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (hw < HW && c < C) v = TT<T>::ld(x + ((size_t)b * HW + hw) * ld + c);
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, hw = hw0 + tx;
        if (c < C && hw < HW) out[((size_t)b * Ctot + c_off + c) * HW + hw] = tile[tx][ty + 8 * k];
    }
}

unsigned grid_for(size_t n, unsigned cap = 2048) {
    size_t g = (n + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

int launch_range_probe(const float* img, size_t n, float mean, float inv_std, float* minmax, hipStream_t s) {
    if (!minmax) return MADM_OK;
    MADM_REQUIRE((reinterpret_cast<uintptr_t>(img) & 15) == 0, "image range probe: image must be 16-byte aligned");
    size_t blocks = (n / 4 + 256 * 8 - 1) / (256 * 8);
    if (blocks > 64) blocks = 64;
    if (blocks < 1) blocks = 1;
    range_probe_kernel<<<(unsigned)blocks, 256, 0, s>>>(img, n, mean, inv_std, minmax);
    return madm_check_launch("range_probe_kernel");
}

}  // namespace

extern "C" {

int madm_image_to_nhwc(int dtype, const float* img, void* out, int B, int C, int H, int W, int Cpad, float mean,
                       float std, float* minmax, void* stream) {
    MADM_REQUIRE(img && out, "image_to_nhwc: null pointer");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && Cpad % epc == 0,
                 "image_to_nhwc: bad dims (C=%d Cpad=%d)", C, Cpad);
    MADM_REQUIRE(std != 0.f, "image_to_nhwc: std == 0");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)B * H * W;
    MADM_DISPATCH_DTYPE(dtype, (image_to_nhwc_kernel<T><<<grid_for(total), 256, 0, s>>>(
                                   img, (T*)out, B, C, H * W, Cpad, mean, 1.0f / std)));
    if (int rc = madm_check_launch("image_to_nhwc_kernel")) return rc;
    return launch_range_probe(img, (size_t)B * C * H * W, mean, 1.0f / std, minmax, s);
}

int madm_image_to_im2col3x3(int dtype, const float* img, void* out, int B, int H, int W, int Kpad, float mean,
                            float std, float* minmax, void* stream) {
    MADM_REQUIRE(img && out, "image_to_im2col3x3: null pointer");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(B > 0 && H > 0 && W > 0 && Kpad >= 27 && Kpad % epc == 0, "image_to_im2col3x3: bad dims");
    MADM_REQUIRE(std != 0.f, "image_to_im2col3x3: std == 0");
    hipStream_t s = (hipStream_t)stream;
    MADM_REQUIRE(Kpad >= 32, "image_to_im2col3x3: Kpad must be at least 32");
    const size_t total = (size_t)B * H * W;
    MADM_REQUIRE(total * Kpad < 0x7fffffffull, "image_to_im2col3x3: tensor too large for 32-bit indexing");
    MADM_DISPATCH_DTYPE(dtype, (image_to_im2col_kernel<T><<<(unsigned)((size_t)B * H * ((W + IM2COL_PX - 1) / IM2COL_PX)), 256, 0, s>>>(
                                   img, (T*)out, B, H, W, Kpad, mean, 1.0f / std)));
    if (int rc = madm_check_launch("image_to_im2col_kernel")) return rc;
    return launch_range_probe(img, (size_t)B * 3 * H * W, mean, 1.0f / std, minmax, s);
}

int madm_stem_conv3x3(int dtype, const float* img, const float* wT, const float* bias, void* out, int ldo, int B, int H,
                      int W, int N, float mean, float std, double* stats, float* minmax, void* stream) {
    MADM_REQUIRE(img && wT && bias && out, "stem_conv3x3: null pointer");
    MADM_REQUIRE(madm_dtype_ok(dtype), "stem_conv3x3: bad dtype");
    MADM_REQUIRE(N == STEM_N, "stem_conv3x3: N = %d, this kernel is the SD VAE stem (N = %d)", N, STEM_N);
    MADM_REQUIRE(B > 0 && H > 0 && W > 0 && ldo >= N && ldo % 4 == 0 && std != 0.f, "stem_conv3x3: bad dims");
    const size_t nblocks = (size_t)B * ((H + STEM_ROWS - 1) / STEM_ROWS) * ((W + STEM_SEG - 1) / STEM_SEG);
    MADM_REQUIRE(nblocks < 0x7fffffffull, "stem_conv3x3: grid too large");
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)nblocks;
    // 16-bit modes: the MFMA kernel (MADM_STEM_KERNEL=1 keeps the exact-f32 FMA kernel for A/B runs); f32: the FMA kernel
    const char* fe = getenv("MADM_STEM_KERNEL");
    const bool fma = dtype == MADM_F32 || (fe && atoi(fe) == 1) || ldo % 8 != 0;
    if (fma) {
        MADM_DISPATCH_DTYPE(dtype, (stem_conv3x3_kernel<T><<<blocks, 256, 0, s>>>(img, wT, bias, (T*)out, ldo, B, H, W, mean,
                                                                                1.0f / std, stats)));
    } else {
        const unsigned mblocks = (unsigned)((size_t)B * ((H + STEM_MFMA_ROWS - 1) / STEM_MFMA_ROWS) * ((W + STEM_SEG - 1) / STEM_SEG));
        if (dtype == MADM_BF16)
            stem_conv3x3_mfma_kernel<bf16_t><<<mblocks, 256, 0, s>>>(img, wT, bias, (bf16_t*)out, ldo, B, H, W, mean, 1.0f / std, stats);
        else
            stem_conv3x3_mfma_kernel<f16_t><<<mblocks, 256, 0, s>>>(img, wT, bias, (f16_t*)out, ldo, B, H, W, mean, 1.0f / std, stats);
    }
    if (int rc = madm_check_launch("stem_conv3x3_kernel")) return rc;
    return launch_range_probe(img, (size_t)B * 3 * H * W, mean, 1.0f / std, minmax, s);
}

int madm_nchw_f32_to_nhwc(int dtype, const float* x, void* out, int B, int C, int HW, int Cpad, void* stream) {
    return madm_image_to_nhwc(dtype, x, out, B, C, HW, 1, Cpad, 0.f, 1.f, nullptr, stream);
}

int madm_latents_add_noise(int dtype, const void* moments, int ldm, float scaling, const float* noise,
                           const float* sqrt_ac, const float* sqrt_1mac, const int64_t* timesteps,
                           float* latents_nchw, void* noisy, int B, int HW, int Cpad, void* stream) {
    MADM_REQUIRE(moments && noise && sqrt_ac && sqrt_1mac && timesteps && noisy, "latents_add_noise: null pointer");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(B > 0 && HW > 0 && ldm >= 4 && Cpad >= 4 && Cpad % epc == 0, "latents_add_noise: bad dims");
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (latents_add_noise_kernel<T><<<grid_for((size_t)B * HW), 256, 0, s>>>(
                                   (const T*)moments, ldm, scaling, noise, sqrt_ac, sqrt_1mac, timesteps,
                                   latents_nchw, (T*)noisy, B, HW, Cpad)));
    return madm_check_launch("latents_add_noise_kernel");
}

int madm_timestep_embedding(int dtype, const int64_t* timesteps, const float* freqs, void* out, int B, int dim,
                            void* stream) {
    MADM_REQUIRE(timesteps && freqs && out && B > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: bad args");
    hipStream_t s = (hipStream_t)stream;
    const int n = B * dim / 2;
    MADM_DISPATCH_DTYPE(dtype, (timestep_embedding_kernel<T><<<(n + 255) / 256, 256, 0, s>>>(timesteps, freqs, (T*)out, B, dim)));
    return madm_check_launch("timestep_embedding_kernel");
}

int madm_silu(int dtype, const void* x, void* y, size_t n, void* stream) {
    MADM_REQUIRE(x && y && n > 0, "silu: bad args");
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (silu_kernel<T><<<grid_for(n), 256, 0, s>>>((const T*)x, (T*)y, n)));
    return madm_check_launch("silu_kernel");
}

int madm_rows_to_f32(int dtype, const void* x, const float* add, float* y, size_t n, void* stream) {
    MADM_REQUIRE(x && y && n > 0, "rows_to_f32: bad args");
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (rows_to_f32_kernel<T><<<grid_for(n), 256, 0, s>>>((const T*)x, add, y, n)));
    return madm_check_launch("rows_to_f32_kernel");
}

int madm_copy_columns(int dtype, const void* src, int lds, void* dst, int ldd, size_t rows, int C, void* stream) {
    MADM_REQUIRE(src && dst && rows > 0 && C > 0 && lds >= C && ldd >= C, "copy_columns: bad args");
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (copy_columns_kernel<T><<<grid_for(rows * C), 256, 0, s>>>((const T*)src, lds, (T*)dst,
                                                                                         ldd, rows, C)));
    return madm_check_launch("copy_columns_kernel");
}

int madm_clamp_f32(const float* x, float* y, size_t n, float lo, float hi, void* stream) {
    MADM_REQUIRE(x && y && n > 0 && lo <= hi, "clamp_f32: bad args");
    clamp_f32_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(x, y, n, lo, hi);
    return madm_check_launch("clamp_f32_kernel");
}

int madm_cast_from_f32(int dtype, const float* x, void* y, size_t n, void* stream) {
    MADM_REQUIRE(x && y && n > 0, "cast_from_f32: bad args");
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (cast_from_f32_kernel<T><<<grid_for(n), 256, 0, s>>>(x, (T*)y, n)));
    return madm_check_launch("cast_from_f32_kernel");
}

int madm_nhwc_to_nchw_f32(int dtype, const void* x, int ld, float* out, int B, int C, int c_off, int Ctot, int HW,
                          void* stream) {
    MADM_REQUIRE(x && out && B > 0 && C > 0 && HW > 0 && ld >= C && c_off >= 0 && c_off + C <= Ctot,
                 "nhwc_to_nchw: bad args");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    MADM_DISPATCH_DTYPE(dtype, (nhwc_to_nchw_kernel<T><<<grid, 256, 0, s>>>((const T*)x, ld, out, C, c_off, Ctot, HW)));
    return madm_check_launch("nhwc_to_nchw_kernel");
}

}  // extern "C"
