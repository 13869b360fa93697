// GroupNorm (stats + apply[+SiLU]) and LayerNorm on channels-last tensors.  HBM-bound
// streaming kernels: 16-byte loads/stores per lane, f32 math, f64 cross-block sums.
#include "common.hpp"

namespace {

// ---- GroupNorm statistics (stand-alone; the conv epilogue normally produces them) ----------
// grid (splits, B).  A block owns rows [r0, r1) of image b.  Threads are laid out as
// (column chunk, row lane); every thread keeps per-channel partial sums of its fixed 16-byte
// column over its rows, adds them into per-channel LDS sums, and the block adds those to the global
// f64 chsums[b][c][2] (sum, sum of squares) with contiguous atomics.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int HW, int C,
                                                       int rows_per_block, double* chsums) {
    constexpr int EPC = TT<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [C][2]
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > HW) r1 = HW;
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) lds[c] = 0.f;
    __syncthreads();

    const int CPR = C / EPC;
    const int cols = CPR < 256 ? CPR : 256;
    const int rowlanes = 256 / cols;
    const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
    if (ty < rowlanes) {
        const T* xb = x + (size_t)b * HW * C;
        for (int q = tx; q < CPR; q += cols) {
            float s[EPC], ss[EPC];
#pragma unroll
            for (int j = 0; j < EPC; ++j) { s[j] = 0.f; ss[j] = 0.f; }
            // four rows per trip, their loads issued together (one load per trip left the kernel at 1.6 .. 2.7 TB/s:
            // tools/exp/bench_norm_bw.py); the sums keep the row order
            for (int r = r0 + ty; r < r1; r += 4 * rowlanes) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = r + u * rowlanes;
                    v[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (rr < r1) v[u] = *reinterpret_cast<const uint4*>(xb + (size_t)rr * C + q * EPC);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {   // (rows past r1 contribute exact zeros)
                    float f[EPC];
                    chunk_to_f32<T>(v[u], f);
#pragma unroll
                    for (int j = 0; j < EPC; ++j) { s[j] += f[j]; ss[j] += f[j] * f[j]; }
                }
            }
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                atomicAdd(&lds[(q * EPC + j) * 2], s[j]);
                atomicAdd(&lds[(q * EPC + j) * 2 + 1], ss[j]);
            }
        }
    }
    __syncthreads();
    double* dst = chsums + (size_t)b * C * 2;
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) atomicAdd(dst + c, (double)lds[c]);
}

// ---- GroupNorm apply (+ optional SiLU) ---------------------------------------------------
// grid (strips, B).  Each block first folds the per-channel sums of the (possibly two-source)
// Ctot-channel tensor into group mean / rstd (f64 arithmetic on the f32 sums), then builds the
// per-channel scale/shift of its source in LDS and streams its rows.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int ldy,
                                                       int HW, int C, int c_off, int Ctot, int G,
                                                       const double* __restrict__ sums1, int C1,
                                                       const double* __restrict__ sums2,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int act,
                                                       const T* __restrict__ res, int ldres, const T* __restrict__ xsrc2) {
    constexpr int EPC = TT<T>::EPC;
    if (xsrc2 && blockIdx.z == 1) {      // both sources of a concatenated input in one launch: grid.z picks the source
        x = xsrc2; c_off = C1; C = Ctot - C1;
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];  // scale[C], shift[C], gmean[G], grstd[G]
    float* scale = lds;
    float* shift = lds + C;
    float* gmean = lds + 2 * C;
    float* grstd = gmean + G;
    const int b = blockIdx.y;
    const int cpg = Ctot / G;
    const double inv_cnt = 1.0 / ((double)HW * (double)cpg);
    const int C2 = Ctot - C1;
    const unsigned CPR = (unsigned)C / EPC;
    const unsigned total = (unsigned)HW * CPR;   // < 2^31 (checked by the launcher)
    const T* xb = x + (size_t)b * HW * C;
    T* yb = y + (size_t)b * HW * ldy + c_off;
    const unsigned idx0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;

    // Everything this block needs from memory is requested up front -- the first PF row chunks (and residuals), the
    // affine parameters, the channel sums -- so the three dependent L2/HBM round trips of the naive order
    // (sums -> gamma/beta -> x) overlap; on the UNet's small tensors this kernel is pure latency.
    constexpr int PF = 4;
    uint4 xv[PF], rv[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const unsigned idx = idx0 + k * stride;
        xv[k] = make_uint4(0, 0, 0, 0);
        rv[k] = make_uint4(0, 0, 0, 0);
        if (idx < total) {
            xv[k] = *reinterpret_cast<const uint4*>(xb + (size_t)idx * EPC);
            if (res) {
                const unsigned r = idx / CPR, q = idx - r * CPR;
                rv[k] = *reinterpret_cast<const uint4*>(res + ((size_t)b * HW + r) * ldres + c_off + q * EPC);
            }
        }
    }
    gamma += c_off;
    beta += c_off;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {   // raw gamma / beta; turned into scale / shift below
        scale[c] = gamma[c];
        shift[c] = beta[c];
    }
    // group fold: 8 lanes per group, each summing every 8th channel of the group with all its loads in flight
    // together, then an xor-shuffle reduction
    for (int g0 = 0; g0 < G; g0 += 32) {
        const int g = g0 + (threadIdx.x >> 3), l = threadIdx.x & 7;
        double s = 0.0, q = 0.0;
        if (g < G) {
            constexpr int UF = 10;   // covers C <= 2560 at 32 groups without a dependent loop
            double sv[UF], qv[UF];
#pragma unroll
            for (int i = 0; i < UF; ++i) {
                const int ch = g * cpg + l + 8 * i;
                sv[i] = 0.0;
                qv[i] = 0.0;
                if (ch < (g + 1) * cpg) {
                    const double* src = (ch < C1) ? sums1 + ((size_t)b * C1 + ch) * 2 : sums2 + ((size_t)b * C2 + (ch - C1)) * 2;
                    sv[i] = src[0];
                    qv[i] = src[1];
                }
            }
#pragma unroll
            for (int i = 0; i < UF; ++i) { s += sv[i]; q += qv[i]; }
            for (int ch = g * cpg + l + 8 * UF; ch < (g + 1) * cpg; ch += 8) {
                const double* src = (ch < C1) ? sums1 + ((size_t)b * C1 + ch) * 2 : sums2 + ((size_t)b * C2 + (ch - C1)) * 2;
                s += src[0];
                q += src[1];
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            s += __shfl_xor(s, o);
            q += __shfl_xor(q, o);
        }
        if (g < G && l == 0) {
            // mean / variance algebra in f64, reciprocal square root in f32 -- exactly as the fused convs fold the same
            // sums (igemm_common.hpp gn_fold_groups): an f64 divide + sqrt is ~100 instructions on the critical path
            const double mean = s * inv_cnt;
            double var = q * inv_cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            gmean[g] = (float)mean;
            grstd[g] = __builtin_amdgcn_rsqf((float)var + eps);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = (c_off + c) / cpg;
        const float sc = grstd[g] * scale[c];
        scale[c] = sc;
        shift[c] = shift[c] - gmean[g] * sc;
    }
    __syncthreads();

    auto emit = [&](unsigned idx, uint4 v, uint4 rvv) {
        const unsigned r = idx / CPR;
        const unsigned q = idx - r * CPR;
        float f[EPC], g[EPC];
        chunk_to_f32<T>(v, f);
#pragma unroll
        for (int j = 0; j < EPC; ++j) f[j] = f[j] * scale[q * EPC + j] + shift[q * EPC + j];
        if (res) {
            chunk_to_f32<T>(rvv, g);
#pragma unroll
            for (int j = 0; j < EPC; ++j) f[j] += g[j];
        }
        act_inplace<EPC>(f, act);
        *reinterpret_cast<uint4*>(yb + (size_t)r * ldy + q * EPC) = f32_to_chunk<T>(f);
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const unsigned idx = idx0 + k * stride;
        if (idx < total) emit(idx, xv[k], rv[k]);
    }
    for (unsigned idx = idx0 + PF * stride; idx < total; idx += stride) {
        const uint4 v = *reinterpret_cast<const uint4*>(xb + (size_t)idx * EPC);
        uint4 rvv = make_uint4(0, 0, 0, 0);
        if (res) {
            const unsigned r = idx / CPR, q = idx - r * CPR;
            rvv = *reinterpret_cast<const uint4*>(res + ((size_t)b * HW + r) * ldres + c_off + q * EPC);
        }
        emit(idx, v, rvv);
    }
}

// ---- GroupNorm finalize: channel sums -> per-(image, channel) affine (scale, shift) ----------------
// y = x * scale + shift == (x - mean_g) * rstd_g * gamma + beta; consumed by the conv that fuses the
// normalisation into its LDS halo load (conv3x3.hip).  grid = B blocks.
__global__ __launch_bounds__(256) void gn_finalize_kernel(int HW, int Ctot, int G, const double* __restrict__ sums1,
                                                          int C1, const double* __restrict__ sums2,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ float gmean[256], grstd[256];
    const int b = blockIdx.x;
    const int cpg = Ctot / G;
    const double inv_cnt = 1.0 / ((double)HW * (double)cpg);
    const int C2 = Ctot - C1;
    for (int g0 = 0; g0 < G; g0 += 32) {
        const int g = g0 + (threadIdx.x >> 3), l = threadIdx.x & 7;
        double s = 0.0, q = 0.0;
        if (g < G) {
            for (int ch = g * cpg + l; ch < (g + 1) * cpg; ch += 8) {
                const double* src = (ch < C1) ? sums1 + ((size_t)b * C1 + ch) * 2 : sums2 + ((size_t)b * C2 + (ch - C1)) * 2;
                s += src[0];
                q += src[1];
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            s += __shfl_xor(s, o);
            q += __shfl_xor(q, o);
        }
        if (g < G && l == 0) {
            // mean / variance algebra in f64, reciprocal square root in f32 -- exactly as the fused convs fold the same
            // sums (igemm_common.hpp gn_fold_groups): an f64 divide + sqrt is ~100 instructions on the critical path
            const double mean = s * inv_cnt;
            double var = q * inv_cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            gmean[g] = (float)mean;
            grstd[g] = __builtin_amdgcn_rsqf((float)var + eps);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Ctot; c += blockDim.x) {
        const int g = c / cpg;
        const float sc = grstd[g] * gamma[c];
        scale[(size_t)b * Ctot + c] = sc;
        shift[(size_t)b * Ctot + c] = beta[c] - gmean[g] * sc;
    }
}

// ---- LayerNorm: one wave per row, row held in registers (C <= 64 * MAXCH * EPC) ---------
template <typename T, int MAXCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, T* __restrict__ y, int M,
                                                        int C, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps) {
    constexpr int EPC = TT<T>::EPC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int CPR = C / EPC;
    const T* xr = x + (size_t)row * C;
    float f[MAXCH][EPC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int q = lane + 64 * i;
        if (q < CPR) {
            uint4 v = *reinterpret_cast<const uint4*>(xr + q * EPC);
            chunk_to_f32<T>(v, f[i]);
#pragma unroll
            for (int j = 0; j < EPC; ++j) s += f[i][j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float v2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int q = lane + 64 * i;
        if (q < CPR) {
#pragma unroll
            for (int j = 0; j < EPC; ++j) { const float d = f[i][j] - mean; v2 += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v2 += __shfl_xor(v2, o);
    const float rstd = 1.0f / sqrtf(v2 / (float)C + eps);
    T* yr = y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int q = lane + 64 * i;
        if (q < CPR) {
            float o[EPC];
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                const int c = q * EPC + j;
                o[j] = (f[i][j] - mean) * rstd * gamma[c] + beta[c];
            }
            *reinterpret_cast<uint4*>(yr + q * EPC) = f32_to_chunk<T>(o);
        }
    }
}

// ---- row softmax: p = softmax(scale * s) over the last dim of f32 [rows][L] -> dtype [rows][ldp] --------
// one wave per row, the row held in registers (L <= 64 * 4 * MAXV); used by the GEMM-based VAE mid-block
// attention (single head, d = 512, L = 4096) where the logits are materialised once in f32.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ sIn, T* __restrict__ pOut,
                                                           int rows, int L, int lds_, int ldp, float scale_log2) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = sIn + (size_t)row * lds_;
    float4 v[MAXV];
    float mx = -INFINITY;
    const int nq = L / 4;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int q = lane + 64 * i;
        if (q < nq) {
            v[i] = *reinterpret_cast<const float4*>(sr + q * 4);
            mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float mb = mx * scale_log2;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int q = lane + 64 * i;
        if (q < nq) {
            v[i].x = __builtin_amdgcn_exp2f(v[i].x * scale_log2 - mb);
            v[i].y = __builtin_amdgcn_exp2f(v[i].y * scale_log2 - mb);
            v[i].z = __builtin_amdgcn_exp2f(v[i].z * scale_log2 - mb);
            v[i].w = __builtin_amdgcn_exp2f(v[i].w * scale_log2 - mb);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    T* pr = pOut + (size_t)row * ldp;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int q = lane + 64 * i;
        if (q < nq) store4<T>(pr + q * 4, f32x4{v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv});
    }
}

}  // namespace

extern "C" {

int madm_softmax_rows(int dtype, const float* s, void* p, int rows, int L, int lds, int ldp, float scale,
                      void* stream) {
    MADM_REQUIRE(s && p && rows > 0 && L > 0 && L % 4 == 0 && lds >= L && ldp >= L && lds % 4 == 0 && ldp % 4 == 0,
                 "softmax_rows: bad args");
    MADM_REQUIRE(L <= 64 * 4 * 16, "softmax_rows: L=%d too long (max 4096)", L);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + 3) / 4));
    const float sl2 = scale * 1.44269504088896340736f;
    MADM_DISPATCH_DTYPE(dtype, (softmax_rows_kernel<T, 16><<<grid, 256, 0, st>>>(s, (T*)p, rows, L, lds, ldp, sl2)));
    return madm_check_launch("softmax_rows_kernel");
}


int madm_groupnorm_stats(int dtype, const void* x, int B, int HW, int C, double* chsums, void* stream) {
    MADM_REQUIRE(x && chsums, "groupnorm_stats: null pointer");
    MADM_REQUIRE(B > 0 && HW > 0 && C > 0, "groupnorm_stats: bad dims B=%d HW=%d C=%d", B, HW, C);
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0, "groupnorm_stats: C=%d must be a multiple of %d", C, epc);
    MADM_REQUIRE((size_t)2 * C * sizeof(float) <= 64 * 1024, "groupnorm_stats: C=%d too large", C);
    // enough blocks to fill the chip, at least 64 rows each
    int splits = (HW + 63) / 64;
    const int maxsplits = (1024 + B - 1) / B;
    if (splits > maxsplits) splits = maxsplits;
    if (splits < 1) splits = 1;
    const int rows_per_block = (HW + splits - 1) / splits;
    splits = (HW + rows_per_block - 1) / rows_per_block;
    dim3 grid((unsigned)splits, (unsigned)B);
    const size_t shm = (size_t)2 * C * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (gn_stats_kernel<T><<<grid, 256, shm, s>>>((const T*)x, HW, C, rows_per_block, chsums)));
    return madm_check_launch("gn_stats_kernel");
}

int madm_groupnorm_apply(int dtype, const void* x, void* y, int ldy, int B, int HW, int C, int c_off, int Ctot,
                         int G, const double* sums1, int C1, const double* sums2, const float* gamma,
                         const float* beta, float eps, int act, const void* residual, int ldres, void* stream) {
    MADM_REQUIRE(x && y && sums1 && gamma && beta, "groupnorm_apply: null pointer");
    MADM_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && Ctot % G == 0 && c_off >= 0 && c_off + C <= Ctot && ldy >= c_off + C,
                 "groupnorm_apply: bad dims");
    MADM_REQUIRE(C1 > 0 && C1 <= Ctot && (C1 == Ctot || sums2), "groupnorm_apply: bad statistics sources (C1=%d Ctot=%d)", C1, Ctot);
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0 && c_off % epc == 0 && ldy % epc == 0, "groupnorm_apply: C/c_off/ldy must be multiples of %d", epc);
    MADM_REQUIRE(act >= 0 && act <= 2 && (!residual || (ldres >= c_off + C && ldres % epc == 0)), "groupnorm_apply: bad act/residual");
    const size_t shm = ((size_t)2 * C + 2 * G) * sizeof(float);
    MADM_REQUIRE(shm <= 64 * 1024, "groupnorm_apply: C=%d too large", C);
    const size_t total = (size_t)HW * (C / epc);
    MADM_REQUIRE(total < 0x7fffffffull, "groupnorm_apply: tensor too large for 32-bit indexing");
    size_t strips = (total + 256 * 4 - 1) / (256 * 4);   // <= 4 chunks per thread: all of them prefetched
    const size_t maxstrips = (size_t)(2048 + B - 1) / B;
    if (strips > maxstrips) strips = maxstrips;
    if (strips < 1) strips = 1;
    dim3 grid((unsigned)strips, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (gn_apply_kernel<T><<<grid, 256, shm, s>>>((const T*)x, (T*)y, ldy, HW, C, c_off, Ctot, G,
                                                                        sums1, C1, sums2, gamma, beta, eps, act,
                                                                        (const T*)residual, ldres, (const T*)nullptr)));
    return madm_check_launch("gn_apply_kernel");
}

int madm_groupnorm_apply_cat(int dtype, const void* x1, const void* x2, void* y, int ldy, int B, int HW, int C1, int C2,
                             int G, const double* sums1, const double* sums2, const float* gamma, const float* beta,
                             float eps, int act, void* stream) {
    MADM_REQUIRE(x1 && x2 && y && sums1 && sums2 && gamma && beta, "groupnorm_apply_cat: null pointer");
    const int Ctot = C1 + C2;
    MADM_REQUIRE(B > 0 && HW > 0 && C1 > 0 && C2 > 0 && G > 0 && Ctot % G == 0 && ldy >= Ctot, "groupnorm_apply_cat: bad dims");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C1 % epc == 0 && C2 % epc == 0 && ldy % epc == 0, "groupnorm_apply_cat: C1/C2/ldy must be multiples of %d", epc);
    MADM_REQUIRE(act >= 0 && act <= 2, "groupnorm_apply_cat: bad act");
    const int Cmax = C1 > C2 ? C1 : C2;
    const size_t shm = ((size_t)2 * Cmax + 2 * G) * sizeof(float);
    MADM_REQUIRE(shm <= 64 * 1024, "groupnorm_apply_cat: C=%d too large", Cmax);
    const size_t total = (size_t)HW * (Cmax / epc);
    MADM_REQUIRE(total < 0x7fffffffull, "groupnorm_apply_cat: tensor too large for 32-bit indexing");
    size_t strips = (total + 256 * 4 - 1) / (256 * 4);
    const size_t maxstrips = (size_t)(2048 + 2 * B - 1) / (2 * B);
    if (strips > maxstrips) strips = maxstrips;
    if (strips < 1) strips = 1;
    dim3 grid((unsigned)strips, (unsigned)B, 2u);
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (gn_apply_kernel<T><<<grid, 256, shm, s>>>((const T*)x1, (T*)y, ldy, HW, C1, 0, Ctot, G, sums1, C1,
                                                                        sums2, gamma, beta, eps, act, (const T*)nullptr, 0,
                                                                        (const T*)x2)));
    return madm_check_launch("gn_apply_kernel");
}

int madm_groupnorm_finalize(int B, int HW, int Ctot, int G, const double* sums1, int C1, const double* sums2,
                            const float* gamma, const float* beta, float eps, float* scale, float* shift,
                            void* stream) {
    MADM_REQUIRE(sums1 && gamma && beta && scale && shift, "groupnorm_finalize: null pointer");
    MADM_REQUIRE(B > 0 && HW > 0 && G > 0 && G <= 256 && Ctot % G == 0 && C1 > 0 && C1 <= Ctot && (C1 == Ctot || sums2),
                 "groupnorm_finalize: bad dims");
    gn_finalize_kernel<<<B, 256, 0, (hipStream_t)stream>>>(HW, Ctot, G, sums1, C1, sums2, gamma, beta, eps, scale, shift);
    return madm_check_launch("gn_finalize_kernel");
}

int madm_layernorm_fwd(int dtype, const void* x, void* y, int M, int C, const float* gamma, const float* beta,
                       float eps, void* stream) {
    MADM_REQUIRE(x && y && gamma && beta, "layernorm: null pointer");
    MADM_REQUIRE(M > 0 && C > 0, "layernorm: bad dims");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0, "layernorm: C=%d must be a multiple of %d", C, epc);
    MADM_REQUIRE(C / epc <= 64 * 5, "layernorm: C=%d too large (max %d)", C, 64 * 5 * epc);
    dim3 grid((unsigned)((M + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (layernorm_kernel<T, 5><<<grid, 256, 0, s>>>((const T*)x, (T*)y, M, C, gamma, beta, eps)));
    return madm_check_launch("layernorm_kernel");
}

}  // extern "C"
