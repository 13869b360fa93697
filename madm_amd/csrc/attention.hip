// Flash-style attention forward for gfx950 (madm_attention_fwd).
//
// Workgroup = NW waves, wave w owns NQ groups of 16 query rows (NQ = 2 on the long self-attention maps: every K / V
// fragment read from LDS then feeds two MFMAs, and the per-tile costs -- next tile's global loads, the LDS hand-over,
// the barrier -- are paid once per 56 MFMAs instead of 28); KV is walked in tiles of NS*16 keys staged in
// LDS.  Everything is kept "query on lane & 15":
//   S^T = K Q^T   (MFMA a = K rows, b = Q rows)  -> lane holds one query, keys 4g..4g+3 per sub-tile
//   row max / sum = in-lane over registers + two xor-shuffles (16, 32)
//   O^T += V^T P^T (MFMA a = V^T, b = P^T)       -> lane holds one query, 4 consecutive d
// P never leaves registers: the accumulator registers of S^T are already the b-operand of the
// second product once the k order of the a-operand (V^T) is permuted to match -- for bf16 that
// operand comes from ds_read_b64_tr_b16 (a 4-key x 16-d block transposed by the LDS), for f32
// from four ds_read_b32.
#include "common.hpp"
#include <type_traits>

namespace {
#ifdef ATTN_STAMPS   // tools/exp/stamps_attn.py: shader-clock stamps per KV tile (block 0 / wave 0)
__device__ unsigned long long g_attn_stamps[1024];
#define A_STAMP(t_, k_) if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (t_) < 128) g_attn_stamps[(t_) * 8 + (k_)] = __builtin_readcyclecounter();
#else
#define A_STAMP(t_, k_)
#endif

struct AttnP {
    const char* q; const char* k; const char* v; char* o;
    int ldq, ldk, ldv, ldo;
    int B, H, Lq, Lk, D;
    float scale_log2;
    unsigned bytes_k, bytes_v;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T, int NKB, int ND, int NW, int NS, int NQ = 1>
struct AttnCfg {
    static constexpr int ES = sizeof(T);
    static constexpr int BQ = NW * 16 * NQ, BKV = NS * 16;
    static constexpr int QK_CH = NKB * 4;            // 16-byte chunks per Q/K row (padded)
    static constexpr int QK_ROWB = NKB * 64 + 16;    // +16 B pad: odd number of 16-B slots
    static constexpr int V_CH = ND * ES;             // chunks per V row: ND*16 elements
    static constexpr int V_ROWB = ND * 16 * ES + 16;
    // small head dims: Q fragments live in registers and K/V tiles are double-buffered in LDS with the next
    // tile's global loads in flight during the MFMAs; d = 512 keeps the single-buffered synchronous form
    static constexpr int KV_TILE_B = BKV * QK_ROWB + BKV * V_ROWB;
    static constexpr int NT = NW * 64;
    static constexpr int KI0 = (BKV * QK_CH + NT - 1) / NT;   // K chunks staged per thread (upper bound)
    static constexpr int VI0 = (BKV * V_CH + NT - 1) / NT;    // V chunks staged per thread (upper bound)
    static constexpr bool PIPE = (BQ * QK_ROWB + 2 * KV_TILE_B) <= 112 * 1024 && (KI0 + VI0) <= 12 && NKB <= 6;
    static constexpr int NBUF = PIPE ? 2 : 1;
    // PIPE: the Q tile is only staged to be read into registers before the KV loop starts; it shares the LDS of the SECOND
    // KV buffer (first written at the end of tile 0) -- two blocks per CU stay resident with 128-query blocks
    static constexpr bool Q_ALIAS = PIPE && BQ * QK_ROWB <= KV_TILE_B;
    static constexpr size_t LDS_BYTES = (Q_ALIAS ? 0 : (size_t)BQ * QK_ROWB) + (size_t)NBUF * KV_TILE_B;
    static constexpr int KI = PIPE ? KI0 : 1, VI = PIPE ? VI0 : 1;
};

typedef unsigned int u32x4a __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2a __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
template <int N> __device__ __forceinline__ void attn_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// waits until at most ``young`` LDS reads issued after the awaited one are outstanding (young is a constant after unrolling)
__device__ __forceinline__ void attn_wait_young(int young) {
    if (young >= 4) attn_wait_lgkm<4>();
    else if (young == 3) attn_wait_lgkm<3>();
    else if (young == 2) attn_wait_lgkm<2>();
    else if (young == 1) attn_wait_lgkm<1>();
    else attn_wait_lgkm<0>();
}
#endif

template <typename T, int NKB, int ND, int NW, int NS, int NQ = 1>
__global__ __launch_bounds__(NW * 64, NQ > 1 ? 2 : 1) void attn_kernel(const AttnP p) {
    using C = AttnCfg<T, NKB, ND, NW, NS, NQ>;
    constexpr int ES = C::ES;
    constexpr int NT = C::NT;
    constexpr unsigned OOB = 0x80000000u;
    // d = 40 in the 16-bit modes (the only head dim below its 16-column tiling: 48): column 40 of the V tile holds 1.0, so
    // O[:, 40] accumulates the row sum in the PV product.  NOTE: that normaliser is the sum of the probabilities AFTER their
    // rounding to the 16-bit MFMA operand type (what the numerator is built from), not the f32 sum the other configurations
    // keep -- numerator and denominator see the same rounded p.  launch_attn refuses any head dim but 40 for it.
    constexpr bool ONES = (ND == 3) && (NKB == 2) && sizeof(T) == 2 && C::PIPE;
    // [r4] 16-bit pipelined configurations: the K fragment reads of S = K Q^T are software-pipelined by hand (inline-asm
    // ds_read_b128, four reads ahead of the MFMAs that consume them, counted lgkmcnt) -- hipcc's schedule waited a full LDS
    // round trip in front of every pair of MFMAs in the first half of the product (in-kernel stamps, d = 40, 128-key tile:
    // 1 490 -> 1 220 clocks for the next tile's global loads + 32 MFMAs)
    constexpr bool MANUAL = C::PIPE && sizeof(T) == 2 && NS % 2 == 0;
    const uint4 vpad = ONES ? make_uint4(std::is_same<T, f16_t>::value ? 0x3C00u : 0x3F80u, 0, 0, 0) : make_uint4(0, 0, 0, 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* KV0 = smem + (C::Q_ALIAS ? 0 : C::BQ * C::QK_ROWB);
    char* Qs = C::Q_ALIAS ? KV0 + C::KV_TILE_B : smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = blockIdx.x * C::BQ;
    const int dch = p.D * ES / 16;  // valid 16-byte chunks per head row

    // ---- stage the Q tile (rows beyond Lq and chunks beyond D are zero) ----
    {
        const char* gq = p.q + ((size_t)(b * p.Lq + q0) * p.ldq + (size_t)h * p.D) * ES;
        const int rows_valid = p.Lq - q0;
        for (int idx = tid; idx < C::BQ * C::QK_CH; idx += NT) {
            const int r = idx / C::QK_CH, c = idx - r * C::QK_CH;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < rows_valid && c < dch)
                v = *reinterpret_cast<const uint4*>(gq + (size_t)r * p.ldq * ES + c * 16);
            *reinterpret_cast<uint4*>(Qs + r * C::QK_ROWB + c * 16) = v;
        }
    }

    // ---- K/V staging state: this thread's chunks of a tile (valid chunks only; the padding chunks of the LDS
    //      rows are zeroed once and never rewritten) ----
    const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)p.k, 0, p.bytes_k, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc((void*)p.v, 0, p.bytes_v, 0x00020000);
    unsigned kgo[C::KI], vgo[C::VI];   // global byte offsets relative to the tile's first key row (OOB = unused slot)
    int klo[C::KI], vlo[C::VI];        // LDS byte offsets inside a K / V tile
    int krow[C::KI], vrow[C::VI];
#pragma unroll
    for (int i = 0; i < (C::PIPE ? C::KI : 0); ++i) {
        const int idx = tid + NT * i;
        const int r = idx / dch, c = idx - r * dch;
        const bool ok = idx < C::BKV * dch;
        krow[i] = ok ? r : (1 << 28);
        kgo[i] = (unsigned)((size_t)r * p.ldk * ES + c * 16);
        klo[i] = r * C::QK_ROWB + c * 16;
    }
#pragma unroll
    for (int i = 0; i < (C::PIPE ? C::VI : 0); ++i) {
        const int idx = tid + NT * i;
        const int r = idx / dch, c = idx - r * dch;
        const bool ok = idx < C::BKV * dch;
        vrow[i] = ok ? r : (1 << 28);
        vgo[i] = (unsigned)((size_t)r * p.ldv * ES + c * 16);
        vlo[i] = r * C::V_ROWB + c * 16;
    }
    for (int nb = 0; nb < (C::Q_ALIAS ? 1 : C::NBUF); ++nb) {   // zero the padding chunks once (aliased buffer 1: after Q is read)
        char* Kb = KV0 + nb * C::KV_TILE_B;
        char* Vb = Kb + C::BKV * C::QK_ROWB;
        for (int idx = tid; idx < C::BKV * C::QK_CH; idx += NT) {
            const int r = idx / C::QK_CH, c = idx - r * C::QK_CH;
            if (c >= dch) *reinterpret_cast<uint4*>(Kb + r * C::QK_ROWB + c * 16) = make_uint4(0, 0, 0, 0);
        }
        for (int idx = tid; idx < C::BKV * C::V_CH; idx += NT) {
            const int r = idx / C::V_CH, c = idx - r * C::V_CH;
            if (c >= dch) *reinterpret_cast<uint4*>(Vb + r * C::V_ROWB + c * 16) = (c == dch) ? vpad : make_uint4(0, 0, 0, 0);
        }
    }
    u32x4a kr[C::KI], vr[C::VI];
#define ATTN_LOAD_KV(k0_)                                                                                  \
    {                                                                                                      \
        const int rows_valid_ = p.Lk - (k0_);                                                              \
        const unsigned kb_ = (unsigned)(((size_t)(b * p.Lk + (k0_)) * p.ldk + (size_t)h * p.D) * ES);      \
        const unsigned vb_ = (unsigned)(((size_t)(b * p.Lk + (k0_)) * p.ldv + (size_t)h * p.D) * ES);      \
        _Pragma("unroll") for (int i = 0; i < C::KI; ++i)                                                  \
            kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsk, krow[i] < rows_valid_ ? kgo[i] + kb_ : OOB, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < C::VI; ++i)                                                  \
            vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsv, vrow[i] < rows_valid_ ? vgo[i] + vb_ : OOB, 0, 0); \
    }
#define ATTN_STORE_KV(buf_)                                                                                \
    {                                                                                                      \
        char* Kb_ = KV0 + (buf_) * C::KV_TILE_B;                                                           \
        char* Vb_ = Kb_ + C::BKV * C::QK_ROWB;                                                             \
        _Pragma("unroll") for (int i = 0; i < C::KI; ++i)                                                  \
            if (krow[i] < C::BKV) *reinterpret_cast<u32x4a*>(Kb_ + klo[i]) = kr[i];                        \
        _Pragma("unroll") for (int i = 0; i < C::VI; ++i)                                                  \
            if (vrow[i] < C::BKV) *reinterpret_cast<u32x4a*>(Vb_ + vlo[i]) = vr[i];                        \
    }

    // single-buffered form (large head dims): straight global -> LDS copy of the valid chunks
#define ATTN_STAGE_DIRECT(k0_)                                                                             \
    {                                                                                                      \
        const int rows_valid_ = p.Lk - (k0_);                                                              \
        char* Kb_ = KV0;                                                                                   \
        char* Vb_ = Kb_ + C::BKV * C::QK_ROWB;                                                             \
        const char* gk_ = p.k + ((size_t)(b * p.Lk + (k0_)) * p.ldk + (size_t)h * p.D) * ES;               \
        const char* gv_ = p.v + ((size_t)(b * p.Lk + (k0_)) * p.ldv + (size_t)h * p.D) * ES;               \
        for (int idx = tid; idx < C::BKV * dch; idx += NT) {                                               \
            const int r = idx / dch, c = idx - r * dch;                                                    \
            uint4 kv_ = make_uint4(0, 0, 0, 0), vv_ = make_uint4(0, 0, 0, 0);                              \
            if (r < rows_valid_) {                                                                         \
                kv_ = *reinterpret_cast<const uint4*>(gk_ + (size_t)r * p.ldk * ES + c * 16);              \
                vv_ = *reinterpret_cast<const uint4*>(gv_ + (size_t)r * p.ldv * ES + c * 16);              \
            }                                                                                              \
            *reinterpret_cast<uint4*>(Kb_ + r * C::QK_ROWB + c * 16) = kv_;                                \
            *reinterpret_cast<uint4*>(Vb_ + r * C::V_ROWB + c * 16) = vv_;                                 \
        }                                                                                                  \
    }

    f32x4 o[NQ][ND];
    float m_run[NQ], l_run[NQ];
#pragma unroll
    for (int g = 0; g < NQ; ++g) {
        m_run[g] = -INFINITY;
        l_run[g] = 0.f;
#pragma unroll
        for (int d = 0; d < ND; ++d) o[g][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (p.Lk + C::BKV - 1) / C::BKV;
    if constexpr (C::PIPE) {
        ATTN_LOAD_KV(0);
        ATTN_STORE_KV(0);
    } else {
        ATTN_STAGE_DIRECT(0);
    }
    __syncthreads();   // Q, padding zeros and tile 0 visible

    uint4 qreg[NQ][C::PIPE ? NKB : 1];
    if constexpr (C::PIPE) {
#pragma unroll
        for (int g = 0; g < NQ; ++g)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
                qreg[g][kb] = *reinterpret_cast<const uint4*>(Qs + ((wave * NQ + g) * 16 + fi) * C::QK_ROWB + (kb * 4 + fg) * 16);
        if constexpr (C::Q_ALIAS) {   // the Q staging area becomes KV buffer 1: restore its zero padding chunks
            __syncthreads();
            char* Kb = KV0 + C::KV_TILE_B;
            char* Vb = Kb + C::BKV * C::QK_ROWB;
            for (int idx = tid; idx < C::BKV * C::QK_CH; idx += NT) {
                const int r = idx / C::QK_CH, c = idx - r * C::QK_CH;
                if (c >= dch) *reinterpret_cast<uint4*>(Kb + r * C::QK_ROWB + c * 16) = make_uint4(0, 0, 0, 0);
            }
            for (int idx = tid; idx < C::BKV * C::V_CH; idx += NT) {
                const int r = idx / C::V_CH, c = idx - r * C::V_CH;
                if (c >= dch) *reinterpret_cast<uint4*>(Vb + r * C::V_ROWB + c * 16) = (c == dch) ? vpad : make_uint4(0, 0, 0, 0);
            }
        }
    }

    for (int t = 0; t < ntiles; ++t) {
        A_STAMP(t, 0);
        const int k0 = t * C::BKV;
        const int cur = C::PIPE ? (t & 1) : 0;
        const bool more = t + 1 < ntiles;
        if constexpr (C::PIPE) {
            if (more) ATTN_LOAD_KV(k0 + C::BKV);   // next tile's loads stay in flight during this tile's MFMAs
            __builtin_amdgcn_sched_barrier(0);
        }
        const char* Ks = KV0 + cur * C::KV_TILE_B;
        const char* Vs = Ks + C::BKV * C::QK_ROWB;

        // ---- S^T = K Q^T for this wave's 16 queries ----
        f32x4 s[NQ][NS];
#pragma unroll
        for (int g = 0; g < NQ; ++g)
#pragma unroll
            for (int st = 0; st < NS; ++st) s[g][st] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (MANUAL) {
#if defined(__HIP_DEVICE_COMPILE__)
            constexpr int NIT = NKB * NS, KD = 4;
            const unsigned kbase = (unsigned)(size_t)Ks + (unsigned)(fi * C::QK_ROWB + fg * 16);
            u32x4a kf[KD + 1];
            attn_wait_lgkm<0>();          // nothing of the compiler's is left on the counter the waits below count on
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < KD && i < NIT; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[i]) : "v"(kbase),
                             "n"(((i % NS) * 16) * C::QK_ROWB + (i / NS) * 64));
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                if (i + KD < NIT)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[(i + KD) % (KD + 1)]) : "v"(kbase),
                                 "n"((((i + KD) % NS) * 16) * C::QK_ROWB + ((i + KD) / NS) * 64));
                attn_wait_young((NIT - 1 - i < KD) ? NIT - 1 - i : KD);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NQ; ++g)
                    mma16<T>(__builtin_bit_cast(uint4, kf[i % (KD + 1)]), qreg[g][i / NS], s[g][i % NS]);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        } else {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            uint4 qf[NQ];
#pragma unroll
            for (int g = 0; g < NQ; ++g) {
                if constexpr (C::PIPE) qf[g] = qreg[g][kb];
                else qf[g] = *reinterpret_cast<const uint4*>(Qs + ((wave * NQ + g) * 16 + fi) * C::QK_ROWB + (kb * 4 + fg) * 16);
            }
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (st * 16 + fi) * C::QK_ROWB + (kb * 4 + fg) * 16);
#pragma unroll
                for (int g = 0; g < NQ; ++g) mma16<T>(kf, qf[g], s[g][st]);
            }
        }
        }
        A_STAMP(t, 1);
        // ---- online softmax (base-2), one query per lane and query group ----
        // [r4] three VALU operations per score instead of six: the running maximum is taken over the RAW scores (scale > 0)
        // and the scale rides on the exponent's FMA, exp2(s * scale - m); with ONES the row sum comes out of the PV product
        // (a ones column in the zero padding of the V tile: O[:, D] accumulates sum(P), rescaled with O) -- no per-score add
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
            float mx = -INFINITY;
            if (k0 + C::BKV <= p.Lk) {   // full tile: no masking (wave-uniform)
#pragma unroll
                for (int st = 0; st < NS; ++st)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[g][st][r]);
            } else {
#pragma unroll
                for (int st = 0; st < NS; ++st)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + st * 16 + fg * 4 + r;
                        const float v = (key < p.Lk) ? s[g][st][r] : -INFINITY;
                        s[g][st][r] = v;
                        mx = fmaxf(mx, v);
                    }
            }
            // row maximum over the four lane groups of a query: v_permlane16/32_swap (VALU) instead of two ds_bpermute round
            // trips -- this exchange sits on the critical path of every KV tile (the exponentials wait for it)
            {
                const unsigned xi = __builtin_bit_cast(unsigned, mx);
                const auto r16 = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
                mx = fmaxf(__builtin_bit_cast(float, r16[0]), __builtin_bit_cast(float, r16[1]));
                const unsigned yi = __builtin_bit_cast(unsigned, mx);
                const auto r32 = __builtin_amdgcn_permlane32_swap(yi, yi, false, false);
                mx = fmaxf(__builtin_bit_cast(float, r32[0]), __builtin_bit_cast(float, r32[1]));
            }
            const float m_new = fmaxf(m_run[g], mx * p.scale_log2);
            const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);
            m_run[g] = m_new;
            const float nm = -m_new;
            float psum = 0.f;
#pragma unroll
            for (int st = 0; st < NS; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[g][st][r], p.scale_log2, nm));
                    s[g][st][r] = e;
                    if constexpr (!ONES) psum += e;
                }
            if constexpr (!ONES) l_run[g] = l_run[g] * alpha + psum;  // per lane-group partial; groups are summed at the end
#pragma unroll
            for (int d = 0; d < ND; ++d) o[g][d] *= alpha;
        }

        A_STAMP(t, 2);
        // ---- O^T += V^T P^T ----
        if constexpr (sizeof(T) == 2) {
            static_assert(sizeof(T) != 2 || NS % 2 == 0, "bf16 path pairs key sub-tiles");
            const int tq = fi >> 2, tp = fi & 3;
            // (the same hand-pipelining of the V fragment reads was built and measured: 1 720 vs 1 610 clocks for this phase --
            //  it takes the P conversion and the exponentials of the second query group out of the MFMA stream, which hipcc
            //  interleaves; the SIMD is issue-bound here, not latency-bound)
#pragma unroll
            for (int u = 0; u < NS / 2; ++u) {
                uint4 pfu[NQ];
#pragma unroll
                for (int g = 0; g < NQ; ++g) {
                    typename TT<T>::vec8 pf;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pf[r] = (T)s[g][2 * u][r];
                        pf[4 + r] = (T)s[g][2 * u + 1][r];
                    }
                    pfu[g] = __builtin_bit_cast(uint4, pf);
                }
                const char* va = Vs + ((2 * u) * 16 + fg * 4 + tq) * C::V_ROWB + tp * 8;
                const char* vb = va + 16 * C::V_ROWB;
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(va + d * 32));
                    s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(vb + d * 32));
                    uint2 a0 = __builtin_bit_cast(uint2, x0), a1 = __builtin_bit_cast(uint2, x1);
                    const uint4 vf = make_uint4(a0.x, a0.y, a1.x, a1.y);
#pragma unroll
                    for (int g = 0; g < NQ; ++g) mma16<T>(vf, pfu[g], o[g][d]);
                }
            }
        } else {
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                uint4 pfu[NQ];
#pragma unroll
                for (int g = 0; g < NQ; ++g)
                    pfu[g] = __builtin_bit_cast(uint4, make_float4(s[g][st][0], s[g][st][1], s[g][st][2], s[g][st][3]));
                const char* vr_ = Vs + (st * 16 + fg * 4) * C::V_ROWB + fi * 4;
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    float4 vf4;
                    vf4.x = *reinterpret_cast<const float*>(vr_ + 0 * C::V_ROWB + d * 64);
                    vf4.y = *reinterpret_cast<const float*>(vr_ + 1 * C::V_ROWB + d * 64);
                    vf4.z = *reinterpret_cast<const float*>(vr_ + 2 * C::V_ROWB + d * 64);
                    vf4.w = *reinterpret_cast<const float*>(vr_ + 3 * C::V_ROWB + d * 64);
#pragma unroll
                    for (int g = 0; g < NQ; ++g) mma16<T>(__builtin_bit_cast(uint4, vf4), pfu[g], o[g][d]);
                }
            }
        }
        A_STAMP(t, 3);
        // ---- hand the next tile over ----
        if constexpr (C::PIPE) {
            __builtin_amdgcn_sched_barrier(0);
            if (more) ATTN_STORE_KV(cur ^ 1);
            A_STAMP(t, 4);
            __syncthreads();
            A_STAMP(t, 5);
        } else {
            __syncthreads();   // everyone done with the single buffer
            if (more) ATTN_STAGE_DIRECT(k0 + C::BKV);
            __syncthreads();
        }
    }
#undef ATTN_LOAD_KV
#undef ATTN_STORE_KV
#undef ATTN_STAGE_DIRECT

    // ---- finish: total row sum over the 4 lane groups, normalise, store 4 consecutive d ----
#pragma unroll
    for (int g = 0; g < NQ; ++g) {
        float l;
        if constexpr (ONES) {   // column 40 of O: tile d = 2, lane group fg = 2, element 0 -> every lane group of the query
            l = __shfl(o[g][2][0], 32 + fi);
        } else {
            l = l_run[g];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
        const float inv = 1.0f / l;
        const int qi = q0 + (wave * NQ + g) * 16 + fi;
        if (qi < p.Lq) {
            T* orow = reinterpret_cast<T*>(p.o) + (size_t)(b * p.Lq + qi) * p.ldo + (size_t)h * p.D;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int dd = d * 16 + fg * 4;
                if (dd < p.D) store4<T>(orow + dd, o[g][d] * inv);
            }
        }
    }
}

template <typename T, int NKB, int ND, int NW, int NS, int NQ = 1>
int launch_attn(const AttnP& p, hipStream_t s) {
    using C = AttnCfg<T, NKB, ND, NW, NS, NQ>;
    auto kern = attn_kernel<T, NKB, ND, NW, NS, NQ>;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = madm_raise_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(C::LDS_BYTES), attr_done, "attention")) return e;
    // the ones-column row sum (ONES in attn_kernel: 16-bit, ND == 3, NKB == 2, pipelined) writes 1.0 into V column 40 and reads
    // the row sum from accumulator column 40: only a head dim of exactly 40 leaves that column free
    if (ND == 3 && NKB == 2 && sizeof(T) == 2 && C::PIPE)
        MADM_REQUIRE(p.D == 40, "attention: the 48-column 16-bit configuration serves head dim 40 only, got %d", p.D);
    dim3 grid((unsigned)((p.Lq + C::BQ - 1) / C::BQ), (unsigned)(p.B * p.H));
    kern<<<grid, NW * 64, C::LDS_BYTES, s>>>(p);
    return madm_check_launch("attn_kernel");
}

}  // namespace

#ifdef ATTN_STAMPS
extern "C" int madm_debug_read_attn_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * n);
}
#endif

// d = 40 at >= 2048 queries: eight-wave workgroups (two query groups per wave) -- round 5, six-slot pipeline, same box, two runs:
// 372.8 / 373.1 -> 375.3 / 374.7 images/s, serial 8.05 -> 8.01 ms (equal in round 4's pipeline).  MADM_ATTN_NW8=0 for the four-wave form.
static const bool g_attn_nw8 = [] { const char* e = getenv("MADM_ATTN_NW8"); return !(e && atoi(e) == 0); }();
static const bool g_attn_nq1 = [] { const char* e = getenv("MADM_ATTN_NQ1"); return e && atoi(e) != 0; }();   // A/B switch

extern "C" int madm_attention_fwd(const madm_attention_args* a, void* stream) {
    MADM_REQUIRE(a && a->q && a->k && a->v && a->o, "attention: null pointer");
    MADM_REQUIRE(a->B > 0 && a->H > 0 && a->Lq > 0 && a->Lk > 0 && a->D > 0, "attention: bad dims");
    const int es = madm_esize(a->dtype);
    MADM_REQUIRE(madm_dtype_ok(a->dtype), "attention: bad dtype");
    MADM_REQUIRE((a->D * es) % 16 == 0, "attention: head dim %d not 16-byte granular", a->D);
    MADM_REQUIRE((a->ldq * es) % 16 == 0 && (a->ldk * es) % 16 == 0 && (a->ldv * es) % 16 == 0 &&
                     (a->ldo * es) % 8 == 0,
                 "attention: row strides must keep 16-byte alignment");
    AttnP p;
    p.q = (const char*)a->q; p.k = (const char*)a->k; p.v = (const char*)a->v; p.o = (char*)a->o;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
    p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk; p.D = a->D;
    p.scale_log2 = a->scale * 1.44269504088896340736f;
    {
        const size_t rows = (size_t)a->B * a->Lk;
        const size_t bk = ((rows - 1) * a->ldk + (size_t)a->H * a->D) * es;
        const size_t bv = ((rows - 1) * a->ldv + (size_t)a->H * a->D) * es;
        MADM_REQUIRE(bk < 0x80000000ull && bv < 0x80000000ull, "attention: K/V views must stay below 2 GiB");
        p.bytes_k = (unsigned)bk; p.bytes_v = (unsigned)bv;
    }
    hipStream_t s = (hipStream_t)stream;
    if (a->dtype == MADM_BF16) {
        switch (a->D) {
            case 40: return a->Lk >= 512 ? (a->Lq >= 2048 && !g_attn_nq1 ? (g_attn_nw8 ? launch_attn<bf16_t, 2, 3, 8, 8, 2>(p, s) : launch_attn<bf16_t, 2, 3, 4, 8, 2>(p, s)) : launch_attn<bf16_t, 2, 3, 4, 8>(p, s))
                                         : launch_attn<bf16_t, 2, 3, 4, 4>(p, s);
            case 64: return launch_attn<bf16_t, 2, 4, 4, 4>(p, s);
            case 80: return a->Lk >= 512 ? launch_attn<bf16_t, 3, 5, 4, 8>(p, s) : launch_attn<bf16_t, 3, 5, 4, 4>(p, s);
            case 160: return launch_attn<bf16_t, 5, 10, 4, 4>(p, s);
            case 512: return launch_attn<bf16_t, 16, 32, 2, 2>(p, s);
            default: break;
        }
    } else if (a->dtype == MADM_F16) {
        switch (a->D) {
            case 40: return a->Lk >= 512 ? (a->Lq >= 2048 && !g_attn_nq1 ? (g_attn_nw8 ? launch_attn<f16_t, 2, 3, 8, 8, 2>(p, s) : launch_attn<f16_t, 2, 3, 4, 8, 2>(p, s)) : launch_attn<f16_t, 2, 3, 4, 8>(p, s))
                                         : launch_attn<f16_t, 2, 3, 4, 4>(p, s);
            case 64: return launch_attn<f16_t, 2, 4, 4, 4>(p, s);
            case 80: return a->Lk >= 512 ? launch_attn<f16_t, 3, 5, 4, 8>(p, s) : launch_attn<f16_t, 3, 5, 4, 4>(p, s);
            case 160: return launch_attn<f16_t, 5, 10, 4, 4>(p, s);
            case 512: return launch_attn<f16_t, 16, 32, 2, 2>(p, s);
            default: break;
        }
    } else {
        switch (a->D) {
            case 40: return launch_attn<float, 3, 3, 4, 4>(p, s);
            case 64: return launch_attn<float, 4, 4, 4, 4>(p, s);
            case 80: return launch_attn<float, 5, 5, 4, 4>(p, s);
            case 160: return launch_attn<float, 10, 10, 4, 4>(p, s);
            case 512: return launch_attn<float, 32, 32, 1, 1>(p, s);
            default: break;
        }
    }
    madm_set_error("attention: head dim %d not instantiated (40, 64, 80, 160, 512)", a->D);
    return MADM_ERR_UNSUPPORTED;
}
