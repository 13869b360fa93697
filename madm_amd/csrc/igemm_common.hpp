// Shared by igemm.hip (gather implicit GEMM) and conv3x3.hip (LDS halo-tile 3x3 conv): launch
// parameters, the common epilogue and the split-K reduction interface.
#pragma once
#include "common.hpp"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct IgemmP {
    const char* in1; const char* in2; const char* w;
    const float* bias; const float* rowvec; const char* residual; char* out; float* ws; double* stats;
    int C1, C2, Ctot, B, IH, IW, OH, OW, KH, KW, stride, pad_t, pad_l, upsample;
    int N, K, M, ldr, ldo, ldrv, epilogue, splitk, tilesN, nk, ld1, ld2, ldw, out_f32;
    unsigned bytes1, bytes2, bytesw;
    // halo-tile 3x3 conv only: fused GroupNorm(+act) of the INPUT, finalized in the kernel from the per-channel f64
    // sums of the (one or two) sources: group g = c / cpg == __umulhi(c, gn_magic)
    const double* gn_sums1; const double* gn_sums2; const float* gn_gamma; const float* gn_beta;
    int gn_G; float gn_eps; unsigned gn_magic; int act;
    // linear layers only: LayerNorm of the INPUT rows folded into the GEMM (madm_conv2d_args.ln_colsum): the kernel reads
    // RAW rows x, w holds gamma-scaled weights W', and  out = rstd_m (x W'^T - mean_m colsum(W')) + bias  with the row
    // sums taken from the A fragments the waves read anyway (K = C: every block walks whole rows)
    const float* ln_cs; float ln_eps;
};

// Row sums of one 16-byte A-fragment chunk (the lane's row, 8 / 4 consecutive k): s += sum x, q += sum x^2 (f32).
template <typename T> __device__ __forceinline__ void ln_accum(const uint4& a, float& s, float& q);
template <> __device__ __forceinline__ void ln_accum<float>(const uint4& a, float& s, float& q) {
    const float4 v = __builtin_bit_cast(float4, a);
    s += (v.x + v.y) + (v.z + v.w);
    q = fmaf(v.x, v.x, q); q = fmaf(v.y, v.y, q); q = fmaf(v.z, v.z, q); q = fmaf(v.w, v.w, q);
}
template <> __device__ __forceinline__ void ln_accum<f16_t>(const uint4& a, float& s, float& q) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const h2 v = __builtin_bit_cast(h2, w[i]);
        s = __builtin_amdgcn_fdot2(v, one, s, false);       // v_dot2c_f32_f16: exact products, f32 accumulation
        q = __builtin_amdgcn_fdot2(v, v, q, false);
    }
}
template <> __device__ __forceinline__ void ln_accum<bf16_t>(const uint4& a, float& s, float& q) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 one = {(__bf16)1.0f, (__bf16)1.0f};
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const b2 v = __builtin_bit_cast(b2, w[i]);
        s = __builtin_amdgcn_fdot2_f32_bf16(v, one, s, false);
        q = __builtin_amdgcn_fdot2_f32_bf16(v, v, q, false);
    }
}

// GroupNorm finalize inside the consumer: gstat[g] = {mean_g, rstd_g} of image b from the per-channel sums.
// All 256 threads: 8 lanes per group, every lane's loads in flight together, xor-shuffle reduction (f64 like
// gn_apply_kernel / gn_finalize_kernel up to the final f32 rsqrt).  The caller synchronises afterwards.
__device__ __forceinline__ void gn_fold_groups(const IgemmP& p, int b, float2* gstat) {
    const int G = p.gn_G, cpg = p.Ctot / G, C2 = p.Ctot - p.C1;
    const double inv_cnt = 1.0 / ((double)(p.IH * p.IW) * (double)cpg);
    for (int g0 = 0; g0 < G; g0 += 32) {
        const int g = g0 + ((int)threadIdx.x >> 3), l = threadIdx.x & 7;
        double s = 0.0, q = 0.0;
        if (g < G) {
            constexpr int UF = 10;
            double sv[UF], qv[UF];
#pragma unroll
            for (int i = 0; i < UF; ++i) {
                const int ch = g * cpg + l + 8 * i;
                sv[i] = 0.0;
                qv[i] = 0.0;
                if (ch < (g + 1) * cpg) {
                    const double* src = (ch < p.C1) ? p.gn_sums1 + ((size_t)b * p.C1 + ch) * 2
                                                    : p.gn_sums2 + ((size_t)b * C2 + (ch - p.C1)) * 2;
                    sv[i] = src[0];
                    qv[i] = src[1];
                }
            }
#pragma unroll
            for (int i = 0; i < UF; ++i) { s += sv[i]; q += qv[i]; }
            for (int ch = g * cpg + l + 8 * UF; ch < (g + 1) * cpg; ch += 8) {
                const double* src = (ch < p.C1) ? p.gn_sums1 + ((size_t)b * p.C1 + ch) * 2
                                                : p.gn_sums2 + ((size_t)b * C2 + (ch - p.C1)) * 2;
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
            // sums and the mean / variance algebra in f64 (cancellation), the reciprocal square root in f32: an f64
            // divide + sqrt is ~100 instructions on this VALU and sat on every block's critical path
            const double mean = s * inv_cnt;
            double var = q * inv_cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            gstat[g] = make_float2((float)mean, __builtin_amdgcn_rsqf((float)var + p.gn_eps));
        }
    }
}

// applies bias / time row / GEGLU / residual, stores 4 (2 for GEGLU) outputs, returns the stored values
template <typename T>
__device__ __forceinline__ f32x4 epilogue_store(const IgemmP& p, int m, int n, f32x4 v) {
    if (p.bias) {
        float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (p.rowvec) {
        const int bi = m / (p.OH * p.OW);
        float4 r = *reinterpret_cast<const float4*>(p.rowvec + (size_t)bi * p.ldrv + n);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    }
    if (p.epilogue == MADM_EPI_GEGLU) {
        float o0 = v[0] * gelu_erf_f(v[1]);
        float o1 = v[2] * gelu_erf_f(v[3]);
        const int col = n >> 1;
        if (p.residual) {
            const T* r = reinterpret_cast<const T*>(p.residual) + (size_t)m * p.ldr + col;
            o0 += TT<T>::ld(r); o1 += TT<T>::ld(r + 1);
        }
        store2<T>(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + col, o0, o1);
        return f32x4{o0, o1, 0.f, 0.f};
    } else {
        if (p.residual) {
            f32x4 r = load4<T>(reinterpret_cast<const T*>(p.residual) + (size_t)m * p.ldr + n);
            v += r;
        }
        if (p.epilogue == MADM_EPI_RELU) {
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        }
        if (p.out_f32) store4<float>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n, v);
        else store4<T>(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n, v);
        return v;
    }
}

// Per-lane constants of a tile epilogue: add[j] = bias[n_j .. n_j+3] (+ the time row of image ``img`` when the whole
// tile lies in one image, img >= 0).  Loaded ONCE per lane before the row loop: inside it the compiler must assume the
// output stores alias them and re-issued the loads for every row -- a full L2 round trip each, 9 000 of the 16 000
// clocks of a 128 x 128 tile's epilogue (in-kernel stamps, tools/exp/stamps_block.py).
template <int NI>
__device__ __forceinline__ void epilogue_consts(const IgemmP& p, int nb, int img, f32x4 (&add)[NI]) {
    // no per-lane condition around the loads (a lane past column N reads the last valid chunk instead -- its values are
    // never stored nor counted): behind exec masks every load got a vmcnt(0) of its own, NI dependent round trips
#pragma unroll
    for (int j = 0; j < NI; ++j) add[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        float4 b[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = nb + 16 * j < p.N ? nb + 16 * j : p.N - 4;
            b[j] = *reinterpret_cast<const float4*>(p.bias + n);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) add[j] = f32x4{b[j].x, b[j].y, b[j].z, b[j].w};
    }
    if (p.rowvec && img >= 0) {
        float4 t[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = nb + 16 * j < p.N ? nb + 16 * j : p.N - 4;
            t[j] = *reinterpret_cast<const float4*>(p.rowvec + (size_t)img * p.ldrv + n);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) { add[j][0] += t[j].x; add[j][1] += t[j].y; add[j][2] += t[j].z; add[j][3] += t[j].w; }
    }
    // The constants must have LANDED before the first output store is issued.  gfx950 counts loads and stores in one
    // vmcnt and they retire out of order with respect to each other, so once a store is in flight the compiler can only
    // wait for an older load with vmcnt(0) -- which also waits for every store.  The loads above sit behind per-column
    // conditions, so without this the "still pending" state survived into the row loop and EVERY row began with a
    // vmcnt(0) behind the previous row's stores: 1 000 .. 1 400 clocks per row, 5 500 of the 18 000 clocks of a
    // 128 x 64 block at K = 320 (in-kernel stamps, tools/exp/stamps_reg.py, round 4).  The empty asm makes the values
    // register-defined here, unconditionally.
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        float a0 = add[j][0], a1 = add[j][1], a2 = add[j][2], a3 = add[j][3];
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        add[j] = f32x4{a0, a1, a2, a3};
    }
}

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4), result in every lane: four rotate-and-add steps on
// the VALU instead of four ds_bpermute round trips
__device__ __forceinline__ float row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));
    return x;
}

// One output pixel row of a wave tile: NI chunks of 4 channels at n = nb + 16 * j.  All residual loads are issued
// first and the NI stores leave back to back, so the (up to) four 32-byte pieces of one 128-byte output line reach
// the L2 together (with a load -> store chain per chunk the L2 evicted half-written lines: 2x HBM write traffic
// measured on the VAE ResNet convs).  ``add`` = epilogue_consts; ``img_rows``: the time row was NOT folded into add
// (tile straddles images) and is fetched per row.  v[j] returns the stored values (fused statistics).
template <typename T, int NI>
__device__ __forceinline__ void epilogue_row(const IgemmP& p, int m, int nb, const f32x4 (&add)[NI], bool img_rows,
                                             f32x4 (&v)[NI]) {
    const float* rv = (p.rowvec && img_rows) ? p.rowvec + (size_t)(m / (p.OH * p.OW)) * p.ldrv + nb : nullptr;
    if (p.epilogue == MADM_EPI_GEGLU) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = nb + 16 * j;
            if (n >= p.N) continue;
            f32x4 t = v[j] + add[j];
            if (rv) {
                const float4 q = *reinterpret_cast<const float4*>(rv + 16 * j);
                t[0] += q.x; t[1] += q.y; t[2] += q.z; t[3] += q.w;
            }
            float o0 = t[0] * gelu_erf_f(t[1]);
            float o1 = t[2] * gelu_erf_f(t[3]);
            const int col = n >> 1;
            if (p.residual) {
                const T* r = reinterpret_cast<const T*>(p.residual) + (size_t)m * p.ldr + col;
                o0 += TT<T>::ld(r); o1 += TT<T>::ld(r + 1);
            }
            store2<T>(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + col, o0, o1);
            v[j] = f32x4{o0, o1, 0.f, 0.f};
        }
        return;
    }
    f32x4 r[NI];
    if (p.residual) {
        const T* rp = reinterpret_cast<const T*>(p.residual) + (size_t)m * p.ldr + nb;
#pragma unroll
        for (int j = 0; j < NI; ++j)
            r[j] = (nb + 16 * j < p.N) ? load4<T>(rp + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        if (nb + 16 * j >= p.N) continue;
        v[j] += add[j];
        if (rv) {
            const float4 t = *reinterpret_cast<const float4*>(rv + 16 * j);
            v[j][0] += t.x; v[j][1] += t.y; v[j][2] += t.z; v[j][3] += t.w;
        }
        if (p.residual) v[j] += r[j];
        if (p.epilogue == MADM_EPI_RELU) {
            v[j][0] = fmaxf(v[j][0], 0.f); v[j][1] = fmaxf(v[j][1], 0.f);
            v[j][2] = fmaxf(v[j][2], 0.f); v[j][3] = fmaxf(v[j][3], 0.f);
        }
    }
    if (p.out_f32) {
        float* op = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + nb;
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (nb + 16 * j < p.N) store4<float>(op + 16 * j, v[j]);
    } else {
        T* op = reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + nb;
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (nb + 16 * j < p.N) store4<T>(op + 16 * j, v[j]);
    }
}

// The MI x NI sub-tiles of one wave: constants, residual, activation for ALL rows first, then all the stores back to
// back -- no load (and so no vmcnt(0), see epilogue_consts) between two stores; the residual pieces of all rows are in
// flight together.  mrow[i] = output row of sub-tile row i or -1 (outside the tensor).  Same operation order per element
// as epilogue_row ((acc + constants) + residual, activation): bit-identical results.  Tiles that straddle images with
// a time row (img_rows) keep the row-by-row form.  acc returns the stored values (fused statistics).
template <typename T, int MI, int NI>
__device__ __forceinline__ void epilogue_tile(const IgemmP& p, const int (&mrow)[MI], int nb, const f32x4 (&add)[NI],
                                              bool img_rows, f32x4 (&acc)[MI][NI]) {
    if ((p.rowvec && img_rows) || (p.epilogue == MADM_EPI_GEGLU && p.residual)) {   // block-uniform, rare
#pragma unroll
        for (int i = 0; i < MI; ++i)
            if (mrow[i] >= 0) epilogue_row<T, NI>(p, mrow[i], nb, add, img_rows, acc[i]);
        return;
    }
    if (p.epilogue == MADM_EPI_GEGLU) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const f32x4 t = acc[i][j] + add[j];
                acc[i][j] = f32x4{t[0] * gelu_erf_f(t[1]), t[2] * gelu_erf_f(t[3]), 0.f, 0.f};
            }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (mrow[i] < 0) continue;
            T* op = reinterpret_cast<T*>(p.out) + (size_t)mrow[i] * p.ldo + (nb >> 1);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (nb + 16 * j < p.N) store2<T>(op + 8 * j, acc[i][j][0], acc[i][j][1]);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] += add[j];
    if (p.residual) {
        // unconditional loads (rows / columns outside the tensor read a valid element instead; their sums are neither
        // stored nor counted): all MI x NI pieces in flight together, one wait
        f32x4 r[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const T* rp = reinterpret_cast<const T*>(p.residual) + (size_t)(mrow[i] < 0 ? 0 : mrow[i]) * p.ldr;
#pragma unroll
            for (int j = 0; j < NI; ++j) r[i][j] = load4<T>(rp + (nb + 16 * j < p.N ? nb + 16 * j : p.N - 4));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] += r[i][j];
    }
    if (p.epilogue == MADM_EPI_RELU) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                f32x4& v = acc[i][j];
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (mrow[i] < 0) continue;
        if (p.out_f32) {
            float* op = reinterpret_cast<float*>(p.out) + (size_t)mrow[i] * p.ldo + nb;
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (nb + 16 * j < p.N) store4<float>(op + 16 * j, acc[i][j]);
        } else {
            T* op = reinterpret_cast<T*>(p.out) + (size_t)mrow[i] * p.ldo + nb;
#pragma unroll
            for (int j = 0; j < NI; ++j)
                if (nb + 16 * j < p.N) store4<T>(op + 16 * j, acc[i][j]);
        }
    }
}

// slow path of the fused GroupNorm statistics: one atomic pair per element (tiles that straddle images)
__device__ __forceinline__ void stats_add_elementwise(const IgemmP& p, int m, int n, f32x4 v) {
    const int bi = m / (p.OH * p.OW);
    double* s = p.stats + ((size_t)bi * p.N + n) * 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        atomicAdd(s + 2 * r, (double)v[r]);
        atomicAdd(s + 2 * r + 1, (double)(v[r] * v[r]));
    }
}


// conv3x3.hip: stride-1 / pad-1 3x3 conv with the input staged as an LDS halo tile (bn = 128 or 64)
template <typename T> int launch_conv3x3_halo(const IgemmP& p, int bn, hipStream_t s);
// conv3x3_dma.hip: the same with the weight tiles moved by LDS-DMA (three-slot ring, single halo buffer)
template <typename T> int launch_conv3x3_halo_dma(const IgemmP& p, int bn, hipStream_t s);
// igemm_apanel.hip (tile 13): linear layers with K = one channel row: resident A panel, streamed weight tiles, optional
// LayerNorm in place; igemm_apanel_bm = rows per panel for this K (0: does not fit)
template <typename T> int launch_igemm_apanel(const IgemmP& p, hipStream_t s);
int igemm_apanel_bm(int K, int esize);
// conv3x3_h16.hip: 16 x 16-pixel patches, 128 x 64 wave tiles, halo and weights by LDS-DMA (maps of at least 16 x 16)
template <typename T> int launch_conv3x3_h16(const IgemmP& p, int bn, hipStream_t s);
