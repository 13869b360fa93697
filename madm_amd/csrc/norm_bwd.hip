// Backward of the stand-alone normalisation kernels of norm.hip (the reference gets them from
// torch autograd, engine/train_loop.py:203-217).  HBM-bound streaming kernels, f32 math, f64 cross-block sums.
//
// GroupNorm(+act), y = act(z), z = xh * gamma + beta, xh = (x - mean_g) * rstd_g  (statistics over the HW * cpg
// elements of a group, possibly across the two sources of a channel concat):
//   dz = dy * act'(z)
//   dbeta[c] = sum dz,  dgamma[c] = sum dz * xh                                 (over images and pixels)
//   dx = rstd_g * (gamma * dz - A_g / cnt - xh * B_g / cnt),  A_g = sum_{c in g, hw} gamma dz,  B_g = ... gamma dz xh
// in two passes over the tensor: "sums" leaves the per-(image, channel) S1 = sum_hw dz and S2 = sum_hw dz * xh in a
// f64 [B][Ctot][2] scratch (same layout as the forward's channel sums), "apply" folds them per group and streams dx.
//
// LayerNorm over the last dim: one wave per row, the row in registers (as the forward kernel), dgamma / dbeta
// accumulated per lane over the wave's rows and added to the f32 outputs with float atomics.
#include "common.hpp"

namespace {

__device__ __forceinline__ float act_grad_f(float z, float dy, int act) {
    if (act == 1) {
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        return dy * sg * (1.0f + z * (1.0f - sg));
    }
    if (act == 2) return z > 0.f ? dy : 0.f;
    return dy;
}

// per-channel {rstd_g, -mean_g * rstd_g} of the channels [c_off, c_off + C) of image b from the forward's channel
// sums; 8 lanes per group.  The caller synchronises.
__device__ __forceinline__ void gn_bwd_fold_stats(float* r, float* mr, int b, int HW, int C, int c_off, int Ctot, int G,
                                                  const double* __restrict__ sums1, int C1,
                                                  const double* __restrict__ sums2, float eps) {
    const int cpg = Ctot / G, C2 = Ctot - C1;
    const double inv_cnt = 1.0 / ((double)HW * (double)cpg);
    for (int g0 = 0; g0 < G; g0 += 32) {
        const int g = g0 + ((int)threadIdx.x >> 3), l = threadIdx.x & 7;
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
        if (g < G) {
            const double mean = s * inv_cnt;
            double var = q * inv_cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            const float rstd = __builtin_amdgcn_rsqf((float)var + eps);   // as the forward folds (norm.hip)
            for (int ch = g * cpg + l; ch < (g + 1) * cpg; ch += 8) {
                const int c = ch - c_off;
                if (c >= 0 && c < C) { r[c] = rstd; mr[c] = -(float)mean * rstd; }
            }
        }
    }
}

// grid (splits, B): per-(image, channel) S1 / S2 of the rows [r0, r1) -> f64 atomics into bsums[b][c_off + c][2]
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_sums_kernel(const T* __restrict__ x, const T* __restrict__ dy, int lddy,
                                                          int HW, int C, int c_off, int Ctot, int G,
                                                          const double* __restrict__ sums1, int C1,
                                                          const double* __restrict__ sums2,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, int act, int rows_per_block, double* bsums) {
    constexpr int EPC = TT<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // r, mr, gamma, beta [C], sums [C][2]
    float* r = lds;
    float* mr = lds + C;
    float* gm = lds + 2 * C;
    float* bt = lds + 3 * C;
    float* acc = lds + 4 * C;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > HW) r1 = HW;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        gm[c] = gamma[c_off + c];
        bt[c] = beta[c_off + c];
        acc[2 * c] = 0.f;
        acc[2 * c + 1] = 0.f;
    }
    gn_bwd_fold_stats(r, mr, b, HW, C, c_off, Ctot, G, sums1, C1, sums2, eps);
    __syncthreads();

    const int CPR = C / EPC;
    const int cols = CPR < 256 ? CPR : 256;
    const int rowlanes = 256 / cols;
    const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
    if (ty < rowlanes) {
        const T* xb = x + (size_t)b * HW * C;
        const T* db = dy + (size_t)b * HW * lddy + c_off;
        for (int q = tx; q < CPR; q += cols) {
            float s1[EPC], s2[EPC], cr[EPC], cm[EPC], cg[EPC], cb[EPC];   // the column's constants live in registers
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                const int c = q * EPC + j;
                s1[j] = 0.f; s2[j] = 0.f;
                cr[j] = r[c]; cm[j] = mr[c]; cg[j] = gm[c]; cb[j] = bt[c];
            }
            // four rows per trip, all eight loads issued together (one row per trip: 1.9 .. 3.2 TB/s for the pair of
            // backward kernels, tools/exp/bench_norm_bw.py); the sums keep the row order
            for (int row = r0 + ty; row < r1; row += 4 * rowlanes) {
                uint4 xv[4], dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = row + u * rowlanes;
                    xv[u] = make_uint4(0u, 0u, 0u, 0u);
                    dv[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (rr < r1) {
                        xv[u] = *reinterpret_cast<const uint4*>(xb + (size_t)rr * C + q * EPC);
                        dv[u] = *reinterpret_cast<const uint4*>(db + (size_t)rr * lddy + q * EPC);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (row + u * rowlanes < r1) {
                        float xf[EPC], df[EPC];
                        chunk_to_f32<T>(xv[u], xf);
                        chunk_to_f32<T>(dv[u], df);
#pragma unroll
                        for (int j = 0; j < EPC; ++j) {
                            const float xh = xf[j] * cr[j] + cm[j];
                            const float dz = act_grad_f(xh * cg[j] + cb[j], df[j], act);
                            s1[j] += dz;
                            s2[j] += dz * xh;
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                atomicAdd(&acc[(q * EPC + j) * 2], s1[j]);
                atomicAdd(&acc[(q * EPC + j) * 2 + 1], s2[j]);
            }
        }
    }
    __syncthreads();
    double* dst = bsums + ((size_t)b * Ctot + c_off) * 2;
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) atomicAdd(dst + c, (double)acc[c]);
}

// grid (strips, B): dx of the source window; block (0, b) also adds image b's share of dgamma / dbeta
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, int lddy,
                                                           T* __restrict__ dx, int HW, int C, int c_off, int Ctot, int G,
                                                           const double* __restrict__ sums1, int C1,
                                                           const double* __restrict__ sums2,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, int act, const double* __restrict__ bsums,
                                                           float* dgamma, float* dbeta, const T* __restrict__ dres,
                                                           int lddres) {
    constexpr int EPC = TT<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // r, mr, gamma, beta, kA, kB [C]
    float* r = lds;
    float* mr = lds + C;
    float* gm = lds + 2 * C;
    float* bt = lds + 3 * C;
    float* kA = lds + 4 * C;
    float* kB = lds + 5 * C;
    const int b = blockIdx.y;
    const int cpg = Ctot / G;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        gm[c] = gamma[c_off + c];
        bt[c] = beta[c_off + c];
    }
    gn_bwd_fold_stats(r, mr, b, HW, C, c_off, Ctot, G, sums1, C1, sums2, eps);
    // A_g / cnt and B_g / cnt: gamma-weighted group sums of S1 / S2 over ALL Ctot channels (both sources)
    const double inv_cnt = 1.0 / ((double)HW * (double)cpg);
    for (int g0 = 0; g0 < G; g0 += 32) {
        const int g = g0 + ((int)threadIdx.x >> 3), l = threadIdx.x & 7;
        double a = 0.0, bb = 0.0;
        if (g < G) {
            for (int ch = g * cpg + l; ch < (g + 1) * cpg; ch += 8) {
                const double* src = bsums + ((size_t)b * Ctot + ch) * 2;
                const double gmc = (double)gamma[ch];
                a += gmc * src[0];
                bb += gmc * src[1];
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            a += __shfl_xor(a, o);
            bb += __shfl_xor(bb, o);
        }
        if (g < G) {
            for (int ch = g * cpg + l; ch < (g + 1) * cpg; ch += 8) {
                const int c = ch - c_off;
                if (c >= 0 && c < C) { kA[c] = (float)(a * inv_cnt); kB[c] = (float)(bb * inv_cnt); }
            }
        }
    }
    if (blockIdx.x == 0 && dgamma) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const double* src = bsums + ((size_t)b * Ctot + c_off + c) * 2;
            unsafeAtomicAdd(dbeta + c_off + c, (float)src[0]);
            unsafeAtomicAdd(dgamma + c_off + c, (float)src[1]);
        }
    }
    __syncthreads();

    // threads = (column chunk, row lane): a thread keeps ONE 16-byte column (its six constants per channel in registers)
    // and walks rows; four rows per trip with all their loads issued before the first store -- with one row per trip every
    // load waited behind the previous row's store (loads and stores share vmcnt on gfx950 and retire out of order with
    // respect to each other: the compiler can only wait with vmcnt(0)) and the kernel ran at half the copy rate
    const int CPR = C / EPC;
    const int cols = CPR < 256 ? CPR : 256;
    const int rowlanes = 256 / cols;
    const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
    if (ty >= rowlanes) return;
    const T* xb = x + (size_t)b * HW * C;
    const T* db = dy + (size_t)b * HW * lddy + c_off;
    T* ob = dx + (size_t)b * HW * C;
    const T* rb = dres ? dres + (size_t)b * HW * lddres + c_off : nullptr;
    const int rstride = (int)gridDim.x * rowlanes;
    for (int q = tx; q < CPR; q += cols) {
        float cr[EPC], cm[EPC], cg[EPC], cb[EPC], ca[EPC], ck[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            const int c = q * EPC + j;
            cr[j] = r[c]; cm[j] = mr[c]; cg[j] = gm[c]; cb[j] = bt[c]; ca[j] = kA[c]; ck[j] = kB[c];
        }
        for (int row = (int)blockIdx.x * rowlanes + ty; row < HW; row += 4 * rstride) {
            uint4 xv[4], dv[4], rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = row + u * rstride;
                xv[u] = make_uint4(0u, 0u, 0u, 0u);
                dv[u] = make_uint4(0u, 0u, 0u, 0u);
                rv[u] = make_uint4(0u, 0u, 0u, 0u);
                if (rr < HW) {
                    xv[u] = *reinterpret_cast<const uint4*>(xb + (size_t)rr * C + q * EPC);
                    dv[u] = *reinterpret_cast<const uint4*>(db + (size_t)rr * lddy + q * EPC);
                    if (rb) rv[u] = *reinterpret_cast<const uint4*>(rb + (size_t)rr * lddres + q * EPC);
                }
            }
            uint4 ov[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float xf[EPC], df[EPC], o[EPC], rf[EPC];
                chunk_to_f32<T>(xv[u], xf);
                chunk_to_f32<T>(dv[u], df);
                chunk_to_f32<T>(rv[u], rf);   // (zeros without a skip-path gradient)
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    const float xh = xf[j] * cr[j] + cm[j];
                    const float dz = act_grad_f(xh * cg[j] + cb[j], df[j], act);
                    o[j] = cr[j] * (cg[j] * dz - ca[j] - xh * ck[j]) + rf[j];
                }
                ov[u] = f32_to_chunk<T>(o);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = row + u * rstride;
                if (rr < HW) *reinterpret_cast<uint4*>(ob + (size_t)rr * C + q * EPC) = ov[u];
            }
        }
    }
}

// ---- LayerNorm backward: one wave per row, rows strided over the grid ---------------------------------------
template <typename T, int MAXCH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            T* __restrict__ dx, int M, int C,
                                                            const float* __restrict__ gamma, float eps, float* dgamma,
                                                            float* dbeta, const T* __restrict__ dres) {
    constexpr int EPC = TT<T>::EPC;
    const int lane = threadIdx.x & 63;
    const int CPR = C / EPC;
    const float invC = 1.0f / (float)C;
    float gm[MAXCH][EPC], ag[MAXCH][EPC], ab[MAXCH][EPC];
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int q = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            gm[i][j] = q < CPR ? gamma[q * EPC + j] : 0.f;
            ag[i][j] = 0.f;
            ab[i][j] = 0.f;
        }
    }
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
        const T* xr = x + (size_t)row * C;
        const T* dr = dy + (size_t)row * C;
        float f[MAXCH][EPC], d[MAXCH][EPC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int q = lane + 64 * i;
#pragma unroll
            for (int j = 0; j < EPC; ++j) { f[i][j] = 0.f; d[i][j] = 0.f; }
            if (q < CPR) {
                chunk_to_f32<T>(*reinterpret_cast<const uint4*>(xr + q * EPC), f[i]);
                chunk_to_f32<T>(*reinterpret_cast<const uint4*>(dr + q * EPC), d[i]);
#pragma unroll
                for (int j = 0; j < EPC; ++j) s += f[i][j];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * invC;
        float v2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            if (lane + 64 * i < CPR) {
#pragma unroll
                for (int j = 0; j < EPC; ++j) { const float t = f[i][j] - mean; v2 += t * t; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v2 += __shfl_xor(v2, o);
        const float rstd = 1.0f / sqrtf(v2 * invC + eps);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            if (lane + 64 * i < CPR) {
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    const float xh = (f[i][j] - mean) * rstd;
                    const float g = gm[i][j] * d[i][j];
                    f[i][j] = xh;
                    m1 += g;
                    m2 += g * xh;
                    ag[i][j] += d[i][j] * xh;
                    ab[i][j] += d[i][j];
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m1 += __shfl_xor(m1, o);
            m2 += __shfl_xor(m2, o);
        }
        m1 *= invC;
        m2 *= invC;
        T* orow = dx + (size_t)row * C;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int q = lane + 64 * i;
            if (q < CPR) {
                float o[EPC], rf[EPC];
#pragma unroll
                for (int j = 0; j < EPC; ++j) rf[j] = 0.f;
                if (dres) chunk_to_f32<T>(*reinterpret_cast<const uint4*>(dres + (size_t)row * C + q * EPC), rf);
#pragma unroll
                for (int j = 0; j < EPC; ++j) o[j] = rstd * (gm[i][j] * d[i][j] - m1 - f[i][j] * m2) + rf[j];
                *reinterpret_cast<uint4*>(orow + q * EPC) = f32_to_chunk<T>(o);
            }
        }
    }
    if (dgamma) {
        // the four waves' column sums meet in LDS first: one global atomic per channel and BLOCK (the grid is capped at
        // 128 blocks) -- one per wave of a 1024-block grid put 4096 adds on every address, 0.5 ms of serialised atomics
        __shared__ float red[2 * 64 * MAXCH * EPC];
        for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) red[c] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int q = lane + 64 * i;
            if (q < CPR) {
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    atomicAdd(&red[q * EPC + j], ag[i][j]);
                    atomicAdd(&red[C + q * EPC + j], ab[i][j]);
                }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            unsafeAtomicAdd(dgamma + c, red[c]);
            unsafeAtomicAdd(dbeta + c, red[C + c]);
        }
    }
}

// GEGLU backward on the interleaved pre-activation rows (value_j, gate_j) the GEGLU GEMM accumulates (pack_geglu_weight
// order): out_j = value_j * gelu(gate_j) ->  dvalue_j = dout_j * gelu(gate_j),  dgate_j = dout_j * value_j * gelu'(gate_j),
// gelu'(g) = Phi(g) + g * phi(g) (exact-erf GELU).  One 16-byte chunk of pre (EPC / 2 pairs) + 8 bytes of dout per thread.
template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const T* __restrict__ pre, const T* __restrict__ dout,
                                                        T* __restrict__ dpre, size_t chunks) {
    constexpr int EPC = TT<T>::EPC, HP = EPC / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (size_t)gridDim.x * blockDim.x) {
        float pv[EPC], o[EPC], dv[HP];
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(pre + i * EPC), pv);
        const T* dp = dout + i * HP;
#pragma unroll
        for (int j = 0; j < HP; ++j) dv[j] = TT<T>::ld(dp + j);
#pragma unroll
        for (int j = 0; j < HP; ++j) {
            const float val = pv[2 * j], g = pv[2 * j + 1];
            const float cdf = 0.5f * (1.0f + erff(g * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * __expf(-0.5f * g * g);
            o[2 * j] = dv[j] * g * cdf;
            o[2 * j + 1] = dv[j] * val * (cdf + g * pdf);
        }
        *reinterpret_cast<uint4*>(dpre + i * EPC) = f32_to_chunk<T>(o);
    }
}

int gn_bwd_check(int dtype, const void* x, const void* dy, int lddy, int B, int HW, int C, int c_off, int Ctot, int G,
                 const double* sums1, int C1, const double* sums2, const float* gamma, const float* beta,
                 const double* bsums, size_t lds_floats) {
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(x && dy && sums1 && gamma && beta && bsums, "groupnorm_bwd: null argument");
    MADM_REQUIRE(B > 0 && HW > 0 && C > 0 && C % epc == 0 && lddy % epc == 0 && c_off % epc == 0,
                 "groupnorm_bwd: C / lddy / c_off must be multiples of %d elements", epc);
    MADM_REQUIRE(G > 0 && Ctot % G == 0 && c_off >= 0 && c_off + C <= Ctot && C1 > 0 && C1 <= Ctot &&
                     (C1 == Ctot || sums2),
                 "groupnorm_bwd: bad channel layout (C %d at %d of %d, G %d, C1 %d)", C, c_off, Ctot, G, C1);
    MADM_REQUIRE(lds_floats * sizeof(float) <= 64 * 1024, "groupnorm_bwd: C = %d too large", C);
    MADM_REQUIRE((long long)HW * (C / epc) < (1ll << 31), "groupnorm_bwd: image too large");
    return MADM_OK;
}

}  // namespace

extern "C" {

int madm_groupnorm_bwd_sums(int dtype, const void* x, const void* dy, int lddy, int B, int HW, int C, int c_off, int Ctot,
                            int G, const double* sums1, int C1, const double* sums2, const float* gamma,
                            const float* beta, float eps, int act, double* bsums, void* stream) {
    const int rc = gn_bwd_check(dtype, x, dy, lddy, B, HW, C, c_off, Ctot, G, sums1, C1, sums2, gamma, beta, bsums,
                                (size_t)6 * C);
    if (rc != MADM_OK) return rc;
    int splits = (HW + 63) / 64;
    const int want = (2048 + B - 1) / B;
    if (splits > want) splits = want;
    if (splits < 1) splits = 1;
    const int rows_per_block = (HW + splits - 1) / splits;
    splits = (HW + rows_per_block - 1) / rows_per_block;
    dim3 grid((unsigned)splits, (unsigned)B);
    const size_t lds = (size_t)6 * C * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (gn_bwd_sums_kernel<T><<<grid, 256, lds, s>>>(
                                   (const T*)x, (const T*)dy, lddy, HW, C, c_off, Ctot, G, sums1, C1, sums2, gamma, beta,
                                   eps, act, rows_per_block, bsums)));
    return madm_check_launch("gn_bwd_sums_kernel");
}

int madm_groupnorm_bwd_apply(int dtype, const void* x, const void* dy, int lddy, void* dx, int B, int HW, int C, int c_off,
                             int Ctot, int G, const double* sums1, int C1, const double* sums2, const float* gamma,
                             const float* beta, float eps, int act, const double* bsums, float* dgamma, float* dbeta,
                             const void* dres, int lddres, void* stream) {
    const int rc = gn_bwd_check(dtype, x, dy, lddy, B, HW, C, c_off, Ctot, G, sums1, C1, sums2, gamma, beta, bsums,
                                (size_t)6 * C);
    if (rc != MADM_OK) return rc;
    MADM_REQUIRE(dx && (!dgamma == !dbeta), "groupnorm_bwd_apply: dx missing or only one of dgamma / dbeta given");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(!dres || lddres % epc == 0, "groupnorm_bwd_apply: lddres must be a multiple of %d elements", epc);
    // a block covers rowlanes rows per step (threads = column chunks x row lanes) and every thread takes four rows per
    // trip: enough blocks that a thread has at most ~4 trips, at most ~4096 blocks per launch
    const int cpr = C / epc;
    const int rowlanes = cpr < 256 ? 256 / cpr : 1;
    int strips = (HW + rowlanes * 16 - 1) / (rowlanes * 16);
    const int want = (4096 + B - 1) / B;
    if (strips > want) strips = want;
    if (strips < 1) strips = 1;
    dim3 grid((unsigned)strips, (unsigned)B);
    const size_t lds = (size_t)6 * C * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (gn_bwd_apply_kernel<T><<<grid, 256, lds, s>>>(
                                   (const T*)x, (const T*)dy, lddy, (T*)dx, HW, C, c_off, Ctot, G, sums1, C1, sums2, gamma,
                                   beta, eps, act, bsums, dgamma, dbeta, (const T*)dres, lddres)));
    return madm_check_launch("gn_bwd_apply_kernel");
}

int madm_geglu_bwd(int dtype, const void* pre, const void* dout, void* dpre, size_t M, int N2, void* stream) {
    MADM_REQUIRE(pre && dout && dpre && M > 0 && N2 > 0, "geglu_bwd: bad argument");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(N2 % epc == 0, "geglu_bwd: 2N = %d must be a multiple of %d", N2, epc);
    const size_t chunks = M * (size_t)(N2 / epc);
    size_t blocks = (chunks + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    MADM_DISPATCH_DTYPE(dtype, (geglu_bwd_kernel<T><<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(
                                   (const T*)pre, (const T*)dout, (T*)dpre, chunks)));
    return madm_check_launch("geglu_bwd_kernel");
}

int madm_layernorm_bwd(int dtype, const void* x, const void* dy, void* dx, int M, int C, const float* gamma, float eps,
                       float* dgamma, float* dbeta, const void* dres, void* stream) {
    MADM_REQUIRE(x && dy && dx && gamma && M > 0 && C > 0, "layernorm_bwd: bad argument");
    MADM_REQUIRE(!dgamma == !dbeta, "layernorm_bwd: dgamma and dbeta go together");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0, "layernorm_bwd: C=%d must be a multiple of %d", C, epc);
    MADM_REQUIRE(C / epc <= 64 * 5, "layernorm_bwd: C=%d too large (max %d)", C, 64 * 5 * epc);
    int blocks = (M + 3) / 4;
    if (blocks > 128) blocks = 128;   // every block adds its column sums once
    hipStream_t s = (hipStream_t)stream;
    if (C / epc <= 64 * 3)   // the UNet's widths in bf16 (320 / 640 / 1280): 3 chunks per lane keep the row state small
        MADM_DISPATCH_DTYPE(dtype, (layernorm_bwd_kernel<T, 3><<<dim3((unsigned)blocks), 256, 0, s>>>(
                                       (const T*)x, (const T*)dy, (T*)dx, M, C, gamma, eps, dgamma, dbeta, (const T*)dres)));
    else
        MADM_DISPATCH_DTYPE(dtype, (layernorm_bwd_kernel<T, 5><<<dim3((unsigned)blocks), 256, 0, s>>>(
                                       (const T*)x, (const T*)dy, (T*)dx, M, C, gamma, eps, dgamma, dbeta, (const T*)dres)));
    return madm_check_launch("layernorm_bwd_kernel");
}

}  // extern "C"
