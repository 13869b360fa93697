// Backward of madm_attention_fwd: dQ, dK, dV of O = softmax(scale Q K^T) V per (image, head).
// The reference gets it from torch autograd through diffusers' Attention (F.scaled_dot_product_attention) under
// losses.backward() (engine/train_loop.py:203-217).
//
// Two launches of ONE kernel template, no atomics, nothing saved by the forward (the caller passes O again):
//   pass "dq"  : block = 64 queries (4 waves x 16), K / V tiles of 64 keys streamed through LDS.  A first sweep
//                recomputes the log-sum-exp of every query row (base 2, scaled logits) and D = rowsum(dO o O) and
//                leaves both in the f32 workspace; the second sweep accumulates dQ.
//   pass "dkv" : block = 64 keys, Q / dO tiles of 64 queries streamed, LSE / D read from the workspace; dK, dV.
// Both passes are the same arithmetic with the roles of the "own" rows (register fragments, one row per lane & 15) and
// the "streamed" rows (LDS tiles) exchanged:
//   G1[s][o] = sum_d A1[s][d] own1[o][d]      (A1, own1) = (K, Q) | (Q, K)          -> S^T | S
//   G2[s][o] = sum_d A2[s][d] own2[o][d]      (A2, own2) = (V, dO) | (dO, V)        -> dP
//   P = exp2(G1 * scale_log2 - LSE[query]),  dS = P o (G2 - D[query])
//   acc1^T[d][o] += sum_s A1^T[d][s] dS[s][o]                                        -> dQ | dK   (x scale at the end)
//   acc2^T[d][o] += sum_s A2^T[d][s] P[s][o]      (dkv only)                         -> dV
// Fragment layouts are those of the forward kernel (attention.hip): the first two products read [rows][d] tiles
// "row = lane & 15, 16-byte chunk = lane >> 4"; P / dS never leave registers and feed the transposed products as the
// b operand, whose a operand (A^T) is read transposed from the same LDS tile -- ds_read_b64_tr_b16 for bf16, four
// ds_read_b32 for f32 -- with the matching permutation of the reduction index.
#include "common.hpp"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct AttnBwdP {
    const char* q; const char* k; const char* v; const char* o; const char* dout;
    char* dq; char* dk; char* dv;
    float* lse; float* dsum;   // f32 [B*H*Lq] each
    int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, H, Lq, Lk, D;
    float scale, scale_log2;
};

template <typename T, int NKB>
struct AttnBwdCfg {
    static constexpr int ES = sizeof(T);
    static constexpr int CH = NKB * 4;            // 16-byte chunks per padded row
    static constexpr int ROWB = NKB * 64 + 16;    // +16 B: odd number of 16-byte slots
    static constexpr int TS = 64;                 // streamed rows per tile
    static constexpr size_t LDS_BYTES = (size_t)2 * TS * ROWB + 2 * TS * sizeof(float);
};

template <typename T, int NKB, int ND, bool DKV>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnBwdP p) {
    using C = AttnBwdCfg<T, NKB>;
    constexpr int ES = C::ES, ROWB = C::ROWB, TS = C::TS, NS = TS / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A1s = smem;
    char* A2s = smem + TS * ROWB;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * TS * ROWB);
    float* d_s = lse_s + TS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / p.H, h = bh - b * p.H;
    const int dch = p.D * ES / 16;
    const int Lown = DKV ? p.Lk : p.Lq, Lstr = DKV ? p.Lq : p.Lk;
    const char* own1p = DKV ? p.k : p.q;
    const char* own2p = DKV ? p.v : p.dout;
    const int ldown1 = DKV ? p.ldk : p.ldq, ldown2 = DKV ? p.ldv : p.lddo;
    const char* s1p = DKV ? p.q : p.k;
    const char* s2p = DKV ? p.dout : p.v;
    const int lds1 = DKV ? p.ldq : p.ldk, lds2 = DKV ? p.lddo : p.ldv;

    const int orow = blockIdx.x * 64 + wave * 16 + fi;   // this lane's own row
    const bool orow_ok = orow < Lown;
    uint4 own1[NKB], own2[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int c = kb * 4 + fg;
        own1[kb] = make_uint4(0, 0, 0, 0);
        own2[kb] = make_uint4(0, 0, 0, 0);
        if (orow_ok && c < dch) {
            own1[kb] = *reinterpret_cast<const uint4*>(own1p + ((size_t)(b * Lown + orow) * ldown1 + (size_t)h * p.D) * ES + c * 16);
            own2[kb] = *reinterpret_cast<const uint4*>(own2p + ((size_t)(b * Lown + orow) * ldown2 + (size_t)h * p.D) * ES + c * 16);
        }
    }

    // stages rows [t0, t0 + 64) of the streamed tensors (zero beyond Lstr and beyond the head dim)
    auto stage = [&](int t0, bool both) {
        const int rows_valid = Lstr - t0;
        for (int idx = tid; idx < TS * C::CH; idx += 256) {
            const int r = idx / C::CH, c = idx - r * C::CH;
            uint4 v1 = make_uint4(0, 0, 0, 0), v2 = make_uint4(0, 0, 0, 0);
            if (r < rows_valid && c < dch) {
                v1 = *reinterpret_cast<const uint4*>(s1p + ((size_t)(b * Lstr + t0 + r) * lds1 + (size_t)h * p.D) * ES + c * 16);
                if (both)
                    v2 = *reinterpret_cast<const uint4*>(s2p + ((size_t)(b * Lstr + t0 + r) * lds2 + (size_t)h * p.D) * ES + c * 16);
            }
            *reinterpret_cast<uint4*>(A1s + r * ROWB + c * 16) = v1;
            if (both) *reinterpret_cast<uint4*>(A2s + r * ROWB + c * 16) = v2;
        }
    };
    // G[st] += A[st rows] . own  (lane: own row fi, streamed rows 4 fg + r of sub-tile st)
    auto gemm_rows = [&](const char* As, const uint4 (&own)[NKB], f32x4 (&g)[NS]) {
#pragma unroll
        for (int st = 0; st < NS; ++st) g[st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const uint4 af = *reinterpret_cast<const uint4*>(As + (st * 16 + fi) * ROWB + (kb * 4 + fg) * 16);
                mma16<T>(af, own[kb], g[st]);
            }
    };
    // acc^T[d][own] += sum_s A^T[d][s] w[s][own] over the 64 streamed rows; w in the G layout
    auto gemm_T = [&](const char* As, const f32x4 (&w)[NS], f32x4 (&acc)[ND]) {
        if constexpr (sizeof(T) == 2) {
            const int tq = fi >> 2, tp = fi & 3;
#pragma unroll
            for (int u = 0; u < NS / 2; ++u) {
                typename TT<T>::vec8 pf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = (T)w[2 * u][r];
                    pf[4 + r] = (T)w[2 * u + 1][r];
                }
                const uint4 pfu = __builtin_bit_cast(uint4, pf);
                const char* va = As + ((2 * u) * 16 + fg * 4 + tq) * ROWB + tp * 8;
                const char* vb = va + 16 * ROWB;
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const s16x4 x0 =
                        __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + d * 32));
                    const s16x4 x1 =
                        __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + d * 32));
                    const uint2 a0 = __builtin_bit_cast(uint2, x0), a1 = __builtin_bit_cast(uint2, x1);
                    mma16<T>(make_uint4(a0.x, a0.y, a1.x, a1.y), pfu, acc[d]);
                }
            }
        } else {
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const float4 pf4 = make_float4(w[st][0], w[st][1], w[st][2], w[st][3]);
                const uint4 pfu = __builtin_bit_cast(uint4, pf4);
                const char* vr = As + (st * 16 + fg * 4) * ROWB + fi * 4;
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    float4 vf;
                    vf.x = *reinterpret_cast<const float*>(vr + 0 * ROWB + d * 64);
                    vf.y = *reinterpret_cast<const float*>(vr + 1 * ROWB + d * 64);
                    vf.z = *reinterpret_cast<const float*>(vr + 2 * ROWB + d * 64);
                    vf.w = *reinterpret_cast<const float*>(vr + 3 * ROWB + d * 64);
                    mma16<T>(__builtin_bit_cast(uint4, vf), pfu, acc[d]);
                }
            }
        }
    };

    const int ntiles = (Lstr + TS - 1) / TS;
    float lse_own = 0.f, d_own = 0.f;
    if constexpr (!DKV) {
        // ---- D = rowsum(dO o O) of the own query ----
        float part = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int c = kb * 4 + fg;
            if (orow_ok && c < dch) {
                const uint4 ov = *reinterpret_cast<const uint4*>(p.o + ((size_t)(b * p.Lq + orow) * p.ldo + (size_t)h * p.D) * ES + c * 16);
                float a[TT<T>::EPC], g[TT<T>::EPC];
                chunk_to_f32<T>(ov, a);
                chunk_to_f32<T>(own2[kb], g);
#pragma unroll
                for (int j = 0; j < TT<T>::EPC; ++j) part += a[j] * g[j];
            }
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        d_own = part;
        // ---- first sweep: log-sum-exp (base 2) of the scaled logits of the own query ----
        float m_run = -INFINITY, l_run = 0.f;
        for (int t = 0; t < ntiles; ++t) {
            stage(t * TS, false);
            __syncthreads();
            f32x4 g1[NS];
            gemm_rows(A1s, own1, g1);
            float mx = -INFINITY;
#pragma unroll
            for (int st = 0; st < NS; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * TS + st * 16 + fg * 4 + r;
                    const float v = key < p.Lk ? g1[st][r] * p.scale_log2 : -INFINITY;
                    g1[st][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            float ps = 0.f;
#pragma unroll
            for (int st = 0; st < NS; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) ps += __builtin_amdgcn_exp2f(g1[st][r] - m_new);
            l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + ps;
            m_run = m_new;
            __syncthreads();
        }
        l_run += __shfl_xor(l_run, 16);
        l_run += __shfl_xor(l_run, 32);
        lse_own = m_run + __log2f(l_run);
        if (fg == 0 && orow_ok) {
            p.lse[(size_t)bh * p.Lq + orow] = lse_own;
            p.dsum[(size_t)bh * p.Lq + orow] = d_own;
        }
    }

    f32x4 acc1[ND], acc2[DKV ? ND : 1];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc1[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < (DKV ? ND : 1); ++d) acc2[d] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < ntiles; ++t) {
        const int t0 = t * TS;
        stage(t0, true);
        if constexpr (DKV) {
            if (tid < TS) {
                const int qi = t0 + tid;
                lse_s[tid] = qi < p.Lq ? p.lse[(size_t)bh * p.Lq + qi] : 0.f;
                d_s[tid] = qi < p.Lq ? p.dsum[(size_t)bh * p.Lq + qi] : 0.f;
            }
        }
        __syncthreads();
        f32x4 g1[NS], g2[NS];
        gemm_rows(A1s, own1, g1);
        gemm_rows(A2s, own2, g2);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            float4 lq = make_float4(lse_own, lse_own, lse_own, lse_own), dq_ = make_float4(d_own, d_own, d_own, d_own);
            if constexpr (DKV) {
                lq = *reinterpret_cast<const float4*>(lse_s + st * 16 + fg * 4);
                dq_ = *reinterpret_cast<const float4*>(d_s + st * 16 + fg * 4);
            }
            const float lv[4] = {lq.x, lq.y, lq.z, lq.w}, dv[4] = {dq_.x, dq_.y, dq_.z, dq_.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = t0 + st * 16 + fg * 4 + r < Lstr;
                const float pr = ok ? __builtin_amdgcn_exp2f(g1[st][r] * p.scale_log2 - lv[r]) : 0.f;
                g1[st][r] = pr;                         // P
                g2[st][r] = pr * (g2[st][r] - dv[r]);   // dS
            }
        }
        gemm_T(A1s, g2, acc1);
        if constexpr (DKV) gemm_T(A2s, g1, acc2);
        __syncthreads();
    }

    if (orow_ok) {
        T* r1 = reinterpret_cast<T*>(DKV ? p.dk : p.dq) + (size_t)(b * Lown + orow) * (DKV ? p.lddk : p.lddq) + (size_t)h * p.D;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int dd = d * 16 + fg * 4;
            if (dd < p.D) store4<T>(r1 + dd, acc1[d] * p.scale);
        }
        if constexpr (DKV) {
            T* r2 = reinterpret_cast<T*>(p.dv) + (size_t)(b * Lown + orow) * p.lddv + (size_t)h * p.D;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int dd = d * 16 + fg * 4;
                if (dd < p.D) store4<T>(r2 + dd, acc2[d]);
            }
        }
    }
}

template <typename T, int NKB, int ND, bool DKV>
int launch_attn_bwd_one(const AttnBwdP& p, hipStream_t s) {
    using C = AttnBwdCfg<T, NKB>;
    auto kern = attn_bwd_kernel<T, NKB, ND, DKV>;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = madm_raise_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(C::LDS_BYTES), attr_done, "attention_bwd")) return e;
    const int Lown = DKV ? p.Lk : p.Lq;
    dim3 grid((unsigned)((Lown + 63) / 64), (unsigned)(p.B * p.H));
    kern<<<grid, 256, C::LDS_BYTES, s>>>(p);
    return madm_check_launch("attn_bwd_kernel");
}

template <typename T, int NKB, int ND>
int launch_attn_bwd(const AttnBwdP& p, hipStream_t s) {
    const int rc = launch_attn_bwd_one<T, NKB, ND, false>(p, s);
    if (rc != MADM_OK) return rc;
    return launch_attn_bwd_one<T, NKB, ND, true>(p, s);
}

}  // namespace

extern "C" size_t madm_attention_bwd_workspace_bytes(const madm_attention_bwd_args* a) {
    if (!a || a->B <= 0 || a->H <= 0 || a->Lq <= 0) return 0;
    return (size_t)2 * a->B * a->H * a->Lq * sizeof(float);
}

extern "C" int madm_attention_bwd(const madm_attention_bwd_args* a, void* stream) {
    MADM_REQUIRE(a && a->q && a->k && a->v && a->o && a->dout && a->dq && a->dk && a->dv, "attention_bwd: null pointer");
    MADM_REQUIRE(a->B > 0 && a->H > 0 && a->Lq > 0 && a->Lk > 0 && a->D > 0, "attention_bwd: bad dims");
    MADM_REQUIRE(madm_dtype_ok(a->dtype), "attention_bwd: bad dtype");
    const int es = madm_esize(a->dtype);
    MADM_REQUIRE((a->D * es) % 16 == 0, "attention_bwd: head dim %d not 16-byte granular", a->D);
    MADM_REQUIRE((a->ldq * es) % 16 == 0 && (a->ldk * es) % 16 == 0 && (a->ldv * es) % 16 == 0 &&
                     (a->ldo * es) % 16 == 0 && (a->lddo * es) % 16 == 0 && (a->lddq * es) % 8 == 0 &&
                     (a->lddk * es) % 8 == 0 && (a->lddv * es) % 8 == 0,
                 "attention_bwd: row strides must keep 16-byte (gradients: 8-byte) alignment");
    MADM_REQUIRE(a->workspace && a->workspace_bytes >= madm_attention_bwd_workspace_bytes(a),
                 "attention_bwd: workspace of %zu bytes needed", madm_attention_bwd_workspace_bytes(a));
    AttnBwdP p;
    p.q = (const char*)a->q; p.k = (const char*)a->k; p.v = (const char*)a->v; p.o = (const char*)a->o;
    p.dout = (const char*)a->dout;
    p.dq = (char*)a->dq; p.dk = (char*)a->dk; p.dv = (char*)a->dv;
    p.lse = (float*)a->workspace;
    p.dsum = p.lse + (size_t)a->B * a->H * a->Lq;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.lddo = a->lddo;
    p.lddq = a->lddq; p.lddk = a->lddk; p.lddv = a->lddv;
    p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk; p.D = a->D;
    p.scale = a->scale;
    p.scale_log2 = a->scale * 1.44269504088896340736f;
    hipStream_t s = (hipStream_t)stream;
    if (a->dtype == MADM_BF16) {
        switch (a->D) {
            case 40: return launch_attn_bwd<bf16_t, 2, 3>(p, s);
            case 64: return launch_attn_bwd<bf16_t, 2, 4>(p, s);
            case 80: return launch_attn_bwd<bf16_t, 3, 5>(p, s);
            case 160: return launch_attn_bwd<bf16_t, 5, 10>(p, s);
            default: break;
        }
    } else if (a->dtype == MADM_F16) {
        switch (a->D) {
            case 40: return launch_attn_bwd<f16_t, 2, 3>(p, s);
            case 64: return launch_attn_bwd<f16_t, 2, 4>(p, s);
            case 80: return launch_attn_bwd<f16_t, 3, 5>(p, s);
            case 160: return launch_attn_bwd<f16_t, 5, 10>(p, s);
            default: break;
        }
    } else {
        switch (a->D) {
            case 40: return launch_attn_bwd<float, 3, 3>(p, s);
            case 64: return launch_attn_bwd<float, 4, 4>(p, s);
            case 80: return launch_attn_bwd<float, 5, 5>(p, s);
            case 160: return launch_attn_bwd<float, 10, 10>(p, s);
            default: break;
        }
    }
    madm_set_error("attention_bwd: head dim %d not instantiated (40, 64, 80, 160)", a->D);
    return MADM_ERR_UNSUPPORTED;
}
