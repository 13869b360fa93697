// A-stationary GEMM for the linear layers whose reduction dim is one channel row (K = C <= 1280) and whose output is
// wide: the fused Q/K/V projection, the cross-attention Q projection and the GEGLU projection of every transformer block
// (diffusers BasicTransformerBlock, driven by modeling/meta_arch/ldm_diffusers.py:454-616).  Tile code 13.
//
//   out[m][n] = sum_k x[m][k] w[n][k] (+ bias, GEGLU),  m = token, K = C
//
// Why: the 64 x 64 / 128 x 64 igemm tiles re-fetch BOTH operand tiles through the CU's load path (64 B / clk) for every
// K step -- 16 .. 24 KB per 0.5 .. 1 MFLOP, twice the MFMA time (igemm.hip, in-kernel stamps) -- and these layers have
// 15 .. 160 column tiles that all re-read the same rows.  Here a workgroup keeps its BM rows resident:
//
//   * the A panel [K / 64][BM][128 B] (80 KB: BM = 128 / 64 / 32 rows at C = 320 / 640 / 1280) is loaded ONCE by LDS-DMA;
//   * the workgroup then walks ``tiles_per_block`` column tiles of 64 channels; only the weight tile of a K step
//     (64 x 128 B = 8 KB) moves: 4-slot LDS-DMA ring, counted vmcnt, one raw s_barrier per step (the igemm_glds skeleton)
//     -> 8 KB per 1 MFLOP (BM = 128): the load path needs half the MFMA time instead of twice;
//   * the LayerNorm in front of the layer (norm1 / norm2 / norm3) is applied IN PLACE on the resident panel (row
//     statistics and x_hat = (x - mean) rstd, once per workgroup; gamma / beta are folded into w / bias by
//     packing.fold_layernorm): no normalisation launch, no per-K-step work, no epilogue correction;
//   * bias values of the block's columns sit in LDS (ordinary global loads inside the DMA loop would make the compiler
//     drain vmcnt(0)); the epilogue of a column tile = bias (+ GEGLU) + stores, then the accumulators restart.
//
// Grid: row panels x column chunks, chunk-major after the XCD remap so that the workgroups of one XCD share weight tiles
// in ITS L2.  Operands as in igemm.hip: [row][8 x 16-byte chunks] images, chunk index XOR (row & 7) applied on the
// source side of the DMA; D = W_frag x A_frag so that a lane holds 4 consecutive channels of one row.
#include <atomic>
#include "igemm_common.hpp"

namespace {

#define AP_ASM(...) asm volatile(__VA_ARGS__)
#if defined(__HIP_DEVICE_COMPILE__)
template <int N> __device__ __forceinline__ void ap_wait_vmcnt() { AP_ASM("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void ap_wait_lgkmcnt() { AP_ASM("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#endif

#ifdef AP_STAMPS   // tools/exp/stamps_apanel.py: shader-clock timeline of wave 0 of workgroup 0
__device__ unsigned long long g_ap_stamps[1024];
#define AP_STAMP(k_) if (blockIdx.x == 0 && threadIdx.x == 0 && (k_) < 1024) g_ap_stamps[(k_)] = __builtin_readcyclecounter();
#else
#define AP_STAMP(k_)
#endif

constexpr int AP_NS = 4;            // weight-tile ring slots (per wave)
constexpr int AP_BN = 64;           // channels per column tile
constexpr int AP_MAX_TPB = 16;      // column tiles per workgroup (bias scratch)

// stores through a buffer descriptor: a lane outside the tensor passes the out-of-range offset and is dropped (branch-free)
template <typename T> __device__ __forceinline__ void ap_store4(__amdgpu_buffer_rsrc_t rs, unsigned off, f32x4 v);
template <> __device__ __forceinline__ void ap_store4<float>(__amdgpu_buffer_rsrc_t rs, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 0);
}
template <> __device__ __forceinline__ void ap_store4<f16_t>(__amdgpu_buffer_rsrc_t rs, unsigned off, f32x4 v) {
    f16x4 o;
    o[0] = (f16_t)v[0]; o[1] = (f16_t)v[1]; o[2] = (f16_t)v[2]; o[3] = (f16_t)v[3];
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rs, off, 0, 0);
}
template <> __device__ __forceinline__ void ap_store4<bf16_t>(__amdgpu_buffer_rsrc_t rs, unsigned off, f32x4 v) {
    bf16x4 o;
    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rs, off, 0, 0);
}
template <typename T> __device__ __forceinline__ void ap_store2(__amdgpu_buffer_rsrc_t rs, unsigned off, float a, float b);
template <> __device__ __forceinline__ void ap_store2<float>(__amdgpu_buffer_rsrc_t rs, unsigned off, float a, float b) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 o = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
    __builtin_amdgcn_raw_buffer_store_b64(o, rs, off, 0, 0);
}
template <> __device__ __forceinline__ void ap_store2<f16_t>(__amdgpu_buffer_rsrc_t rs, unsigned off, float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 o = {(_Float16)a, (_Float16)b};
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rs, off, 0, 0);
}
template <> __device__ __forceinline__ void ap_store2<bf16_t>(__amdgpu_buffer_rsrc_t rs, unsigned off, float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 o = {(__bf16)a, (__bf16)b};
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rs, off, 0, 0);
}

// WM = waves along the rows: 2 -> 2 x 2 waves, wave tile (BM / 2) x 32; 1 -> 1 x 4 waves, wave tile BM x 16.
// After the prologue the waves never synchronise: the panel is read-only and every wave streams the weight rows of ITS
// columns into a private ring (with WM = 2 the two waves of a column half fetch the same rows: twice the weight bytes on
// the load path, still half of what a 64 x 64 tile moves per FLOP, and no barrier in the loop).
// Built WITHOUT packed-FP32 VALU ops (device pass only; see csrc/Makefile's note for what was observed with them).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AP_PKCHECK)
#define AP_NO_PACKED_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define AP_NO_PACKED_F32
#endif

#ifdef AP_PKCHECK   // tools/exp/pkf32_check.py: the kernel WITH packed-FP32 ops, every normalised element re-derived by scalar
                    // v_sub_f32 / v_mul_f32 from the same registers; mismatches are recorded (16 words each, first 60)
__device__ unsigned g_ap_pk[1024];
#endif

template <typename T, int BM, int WM>
__global__ __launch_bounds__(256, 2) AP_NO_PACKED_F32 void igemm_apanel_kernel(const IgemmP p, int tpb, int nchunks) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int EPC = TT<T>::EPC;
    constexpr int WN = 4 / WM;
    constexpr int WCOLS = AP_BN / WN;                    // columns per wave: 32 / 16
    constexpr int MI = BM / (16 * WM), NI = WCOLS / 16;
    constexpr int WP = WCOLS / 8;                        // DMA pieces (8 rows x 128 B) per wave and step: 4 / 2
    constexpr unsigned SLOT = WCOLS * 128u;              // bytes per ring slot
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) uint4 apsmem[];   // [KC][BM][8] panel | [4 waves][NS][WCOLS][8] | bias
    typedef __attribute__((address_space(3))) char lds_char;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (WM == 2) ? (wave >> 1) : 0, wn = (WM == 2) ? (wave & 1) : wave;
    const int KC = p.nk;                                  // 128-byte chunks per row
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int P = (p.M + BM - 1) / BM;
    const int chunk = bid / P, panel = bid - chunk * P;
    const int m0 = panel * BM;
    const int tilesN = (p.N + AP_BN - 1) / AP_BN;
    const int t0 = chunk * tpb;
    int nt = tilesN - t0; nt = nt > tpb ? tpb : nt;       // column tiles of this workgroup
    const int S = nt * KC;                                // K steps in all

    AP_STAMP(0);
    lds_char* const lds0 = (lds_char*)apsmem;
    const unsigned panel_bytes = (unsigned)KC * BM * 128u;
    const unsigned ring_bytes = 4u * AP_NS * SLOT;
    lds_char* const ring0 = lds0 + panel_bytes + (unsigned)wave * (AP_NS * SLOT);   // this wave's ring
    float* const sbias = reinterpret_cast<float*>(apsmem) + (panel_bytes + ring_bytes) / 4;

    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.bytesw, 0x00020000);
    const int ocols = (p.epilogue == MADM_EPI_GEGLU) ? p.N / 2 : p.N;
    const unsigned oes = p.out_f32 ? 4u : (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.out, 0, (unsigned)(((size_t)(p.M - 1) * p.ldo + ocols) * oes), 0x00020000);
    const int rsub = lane >> 3, cpos = lane & 7;
    const unsigned swz = (unsigned)((cpos ^ rsub) * EPC * (int)sizeof(T));   // source-side XOR swizzle (row & 7 == rsub)

    // ---- the bias values of this workgroup's columns -> LDS (before any DMA is in flight) ----
    for (int i = tid; i < nt * AP_BN; i += 256) {
        const int n = t0 * AP_BN + i;
        sbias[i] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }

    // ---- A panel: KC x BM / 8 pieces of 8 rows x 128 B; wave w takes row blocks w, w + 4, ... and every chunk of them ----
#pragma unroll
    for (int rb = 0; rb < BM / 32; ++rb) {
        const int row = (wave + 4 * rb) * 8 + rsub;
        const int m = m0 + row;
        const unsigned base = (m < p.M) ? (unsigned)m * (unsigned)p.ld1 * (unsigned)sizeof(T) + swz : OOB;
        for (int c = 0; c < KC; ++c) {
            lds_char* dst = lds0 + ((unsigned)c * BM + (unsigned)(wave + 4 * rb) * 8u) * 128u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, base, (unsigned)c * 128u, 0, 0);
        }
    }

    // ---- weight rows of this wave's columns: WP pieces per step; (lt, lc) = column tile / chunk of the NEXT request ----
    int lt = 0, lc = 0;
    unsigned wvoff[WP];
    auto set_wvoff = [&](int t) {
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int n = (t0 + t) * AP_BN + wn * WCOLS + j * 8 + rsub;
            wvoff[j] = (n < p.N) ? (unsigned)n * (unsigned)p.ldw * (unsigned)sizeof(T) + swz : OOB;
        }
    };
    set_wvoff(0);
#define AP_LOAD_W(slot)                                                                                          \
    {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < WP; ++j) {                                                         \
            lds_char* dst = ring0 + (unsigned)(slot) * SLOT + (unsigned)j * 1024u;                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, wvoff[j], (unsigned)lc * 128u, 0, 0);         \
        }                                                                                                        \
        if (++lc == KC) { lc = 0; ++lt; set_wvoff(lt); }                                                         \
    }
#pragma unroll
    for (int u = 0; u < AP_NS - 1; ++u)
        if (u < S) AP_LOAD_W(u);

    // ---- every wave's panel pieces landed -> barrier; optional LayerNorm in place ----
    AP_STAMP(1);
    AP_ASM("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    AP_STAMP(2);
    __builtin_amdgcn_s_barrier();
    AP_STAMP(3);
    if (p.ln_cs) {
        // 16 lanes per row, 4 rows per wave and round: every lane keeps its 16-byte pieces in registers between the
        // statistics and the write-back.  Piece q of a row = chunk q / 8, position q % 8 (order does not matter here).
        const int l16 = lane & 15, rq = lane >> 4;
        const int npieces = KC * 8;
        constexpr int PMAX = 10;                                  // a row is at most 2560 B = 160 pieces (80 KB / 32 rows)
        const float inv_k = 1.0f / (float)p.K;
        for (int r0 = wave * 4 + rq; r0 < BM; r0 += 16) {
            uint4 pc[PMAX];
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int u = 0; u < PMAX; ++u) {
                const int pi = l16 + 16 * u;
                if (pi < npieces) {
                    pc[u] = apsmem[((pi >> 3) * BM + r0) * 8 + (pi & 7)];
                    ln_accum<T>(pc[u], s, q);
                }
            }
            s = row16_sum(s);
            q = row16_sum(q);
            const float mean = s * inv_k;
            float var = q * inv_k - mean * mean;
            var = var < 0.f ? 0.f : var;
            const float rstd = 1.0f / sqrtf(var + p.ln_eps);
#pragma unroll
            for (int u = 0; u < PMAX; ++u) {
                const int pi = l16 + 16 * u;
                if (pi < npieces) {
                    float f[EPC];
                    chunk_to_f32<T>(pc[u], f);
#ifdef AP_PKCHECK
                    float f0[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) f0[e] = f[e];
#ifdef AP_PKNOP     // experiment: wait states between the EXEC write that opens this branch and the first packed op
                    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#endif
#endif
#pragma unroll
                    for (int e = 0; e < EPC; ++e) f[e] = (f[e] - mean) * rstd;
#ifdef AP_PKCHECK
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        float g;
                        asm volatile("v_sub_f32 %0, %1, %2\n\ts_nop 1\n\tv_mul_f32 %0, %0, %3" : "=&v"(g) : "v"(f0[e]), "v"(mean), "v"(rstd));
                        if (__builtin_bit_cast(unsigned, g) != __builtin_bit_cast(unsigned, f[e])) {
                            const unsigned k = atomicAdd(&g_ap_pk[0], 1u);
                            if (k < 60) {
                                unsigned* r = g_ap_pk + 16 + k * 16;
                                const unsigned long long ex = __builtin_amdgcn_read_exec();
                                r[0] = blockIdx.x; r[1] = tid; r[2] = (unsigned)r0; r[3] = (unsigned)u; r[4] = (unsigned)e;
                                r[5] = __builtin_bit_cast(unsigned, f0[e]); r[6] = __builtin_bit_cast(unsigned, mean);
                                r[7] = __builtin_bit_cast(unsigned, rstd); r[8] = __builtin_bit_cast(unsigned, f[e]);
                                r[9] = __builtin_bit_cast(unsigned, g); r[10] = (unsigned)ex; r[11] = (unsigned)(ex >> 32);
                                r[12] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
                                r[13] = (unsigned)npieces; r[14] = (unsigned)BM; r[15] = 0xabcd0000u + (unsigned)EPC;
                            }
                        }
                    }
#endif
                    apsmem[((pi >> 3) * BM + r0) * 8 + (pi & 7)] = f32_to_chunk<T>(f);
                }
            }
        }
        AP_ASM("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- main loop: no barrier, the wave's own counted vmcnt orders its ring ----
    AP_STAMP(4);
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fg = lane >> 4, fsw = lane & 7;
    unsigned aA[2], aB[2];
    {
        const unsigned base = (unsigned)(size_t)lds0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = (fg + 4 * kk) ^ fsw;
            aA[kk] = base + (unsigned)(((wm * (BM / WM) + frow) * 8 + c) * 16);
            aB[kk] = base + panel_bytes + (unsigned)wave * (AP_NS * SLOT) + (unsigned)((frow * 8 + c) * 16);
        }
    }
    const unsigned bias_base = (unsigned)(size_t)lds0 + panel_bytes + ring_bytes;
    int ct = 0, cc = 0;       // column tile / chunk of the current step
    int landed = 0;           // the tiles of the next ``landed`` steps are known to have landed: no wait
    for (int s = 0; s < S; ++s) {
        // vmcnt counts this wave's DMA pieces AND its stores, and stores retire out of order with respect to loads (a
        // count that assumes "the E youngest operations are the stores" is satisfied early when they complete first:
        // observed as rare stale weight tiles).  So counted waits only ever assume LOADS younger than the awaited tile --
        // outstanding stores then merely make the wait longer -- and an epilogue drains its tiles before it stores.
        if (landed > 0) --landed;
        else {
            const int ahead = (S - 1 - s < AP_NS - 2) ? S - 1 - s : AP_NS - 2;   // tiles requested after tile s
            const int young = ahead * WP;
            if (young >= 8) ap_wait_vmcnt<8>();
            else if (young == 4) ap_wait_vmcnt<4>();
            else if (young == 2) ap_wait_vmcnt<2>();
            else ap_wait_vmcnt<0>();
        }
        AP_STAMP(8 + s * 4);
        AP_STAMP(8 + s * 4 + 1);
        if (s + AP_NS - 1 < S) AP_LOAD_W((s + AP_NS - 1) & (AP_NS - 1));
        AP_STAMP(8 + s * 4 + 2);
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned pofs = (unsigned)cc * (unsigned)(BM * 128);
            const unsigned rofs = (unsigned)(s & (AP_NS - 1)) * SLOT;
            u32x4 af[2][MI], wf[2][NI];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned pa = aA[kk] + pofs, pb = aB[kk] + rofs;
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    AP_ASM("ds_read_b128 %0, %1 offset:%2" : "=v"(af[kk][i]) : "v"(pa), "n"(i * 2048));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    AP_ASM("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[kk][j]) : "v"(pb), "n"(j * 2048));
            }
            ap_wait_lgkmcnt<MI + NI>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    mma16<T>(__builtin_bit_cast(uint4, wf[0][j]), __builtin_bit_cast(uint4, af[0][i]), acc[i][j]);
            ap_wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    mma16<T>(__builtin_bit_cast(uint4, wf[1][j]), __builtin_bit_cast(uint4, af[1][i]), acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        AP_STAMP(8 + s * 4 + 3);
        if (++cc == KC) {
            // ---- epilogue of column tile ct: bias (+ GEGLU), stores; the lane holds row i * 16 + frow, channels 4 fg .. + 3 ----
            cc = 0;
            const int nloc = ct * AP_BN + wn * WCOLS + fg * 4;          // column inside the block's range
            const int nb = t0 * AP_BN + nloc;
            f32x4 bv[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const unsigned ba = bias_base + (unsigned)(nloc + 16 * j) * 4u;
                AP_ASM("ds_read_b128 %0, %1" : "=v"(bv[j]) : "v"(ba));
            }
            ap_wait_lgkmcnt<0>();
            const bool geglu = p.epilogue == MADM_EPI_GEGLU;
            // values first (the tiles requested for the next steps keep landing meanwhile), then the drain, then the stores
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    f32x4 v = acc[i][j] + bv[j];
                    if (geglu) {
                        v[0] = v[0] * gelu_erf_f(v[1]);
                        v[1] = v[2] * gelu_erf_f(v[3]);
                    } else if (p.epilogue == MADM_EPI_RELU) {
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                    }
                    acc[i][j] = v;
                }
            ap_wait_vmcnt<0>();
            landed = (S - 1 - s < AP_NS - 1) ? S - 1 - s : AP_NS - 1;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * (BM / WM) + i * 16 + frow;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int n = nb + 16 * j;
                    const f32x4 v = acc[i][j];
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const bool ok = m < p.M && n < p.N;
                    if (geglu) ap_store2<T>(rso, ok ? (unsigned)(((size_t)m * p.ldo + (n >> 1)) * sizeof(T)) : OOB, v[0], v[1]);
                    else if (p.out_f32) ap_store4<float>(rso, ok ? (unsigned)(((size_t)m * p.ldo + n) * 4u) : OOB, v);
                    else ap_store4<T>(rso, ok ? (unsigned)(((size_t)m * p.ldo + n) * sizeof(T)) : OOB, v);
                }
            }
            ++ct;
        }
    }
    AP_STAMP(5);
#undef AP_LOAD_W
#endif
}

template <typename T, int BM, int WM>
int launch_apanel_bm(const IgemmP& p, hipStream_t s) {
    const int KC = p.nk;
    const size_t lds = (size_t)KC * BM * 128 + (size_t)4 * AP_NS * (AP_BN / (4 / WM)) * 128 +
                       (size_t)AP_MAX_TPB * AP_BN * sizeof(float);
    static const size_t pad = [] { const char* e = getenv("MADM_APANEL_LDS_PAD"); return e ? (size_t)atol(e) : (size_t)0; }();
    auto kern = igemm_apanel_kernel<T, BM, WM>;
    static std::atomic<uint64_t> attr_set{0};   // per device (the attribute is)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) {
            madm_set_error("igemm (A panel): cannot raise dynamic LDS: %s", hipGetErrorString(e));
            return MADM_ERR_LAUNCH;
        }
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    const int P = (p.M + BM - 1) / BM;
    const int tilesN = (p.N + AP_BN - 1) / AP_BN;
    // one round of workgroups (one per CU: the panel takes half the LDS), at most AP_MAX_TPB column tiles each
    static const int target = [] { const char* e = getenv("MADM_APANEL_BLOCKS"); return e ? atoi(e) : 512; }();
    int nsplit = (target + P - 1) / P;
    if (nsplit > tilesN) nsplit = tilesN;
    if (nsplit < 1) nsplit = 1;
    int tpb = (tilesN + nsplit - 1) / nsplit;
    if (tpb > AP_MAX_TPB) tpb = AP_MAX_TPB;
    const int nchunks = (tilesN + tpb - 1) / tpb;
    kern<<<dim3((unsigned)(P * nchunks)), 256, lds + pad, s>>>(p, tpb, nchunks);
    return madm_check_launch("igemm_apanel_kernel");
}

}  // namespace

// rows the panel may hold for this reduction width, or 0 when the layer does not fit (row bytes x BM <= 80 KB)
int igemm_apanel_bm(int K, int esize) {
    static const int force = [] { const char* e = getenv("MADM_APANEL_BM"); return e ? atoi(e) : 0; }();
    const size_t row = (size_t)K * esize;
    if (force && row * force <= 96 * 1024) return force;
    // 40 KB panels where the rows allow it: two workgroups per CU (72 KB each with the rings) overlap each other's
    // prologue / epilogue / DMA issue -- measured on MI355X, f16: GEGLU 8192 x 320 -> 2560  36.0 us against 40.8 us with
    // one 128-row panel per CU (and 40.3 us for the 128 x 64 igemm tile)
    if (row * 64 <= 40 * 1024) return 64;
    if (row * 32 <= 40 * 1024) return 32;
    if (row * 32 <= 80 * 1024) return 32;
    return 0;
}

template <typename T>
int launch_igemm_apanel(const IgemmP& p, hipStream_t s) {
    const int bm = igemm_apanel_bm(p.K, (int)sizeof(T));
    if (bm == 128) return launch_apanel_bm<T, 128, 2>(p, s);
    if (bm == 64) return launch_apanel_bm<T, 64, 1>(p, s);
    if (bm == 32) return launch_apanel_bm<T, 32, 1>(p, s);
    madm_set_error("igemm (A panel): K = %d does not fit the panel", p.K);
    return MADM_ERR_UNSUPPORTED;
}

#ifdef AP_PKCHECK
extern "C" int madm_debug_read_ap_pkcheck(unsigned* host, int n, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ap_pk), sizeof(unsigned) * n);
    if (reset) {
        static unsigned zeros[1024];
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ap_pk), zeros, sizeof(zeros));
    }
    return rc;
}
#endif

#ifdef AP_STAMPS
extern "C" int madm_debug_read_ap_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ap_stamps), sizeof(unsigned long long) * n);
}
#endif

template int launch_igemm_apanel<float>(const IgemmP&, hipStream_t);
template int launch_igemm_apanel<bf16_t>(const IgemmP&, hipStream_t);
template int launch_igemm_apanel<f16_t>(const IgemmP&, hipStream_t);
