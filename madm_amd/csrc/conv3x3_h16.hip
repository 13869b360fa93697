// 16 x 16-pixel halo-tile 3x3 convolution (tile code 12): the round-2 successor of conv3x3_dma.hip for the large maps of
// the VAE encoder / decoder and the segmentation head (stride 1, pad 1, >= 16 x 16 maps).
//
// What changes against the 8 x 16 kernel, and why (in-kernel stamps of round 1: a tap step cost ~1650 clocks for 1024
// clocks of MFMA, a block spent 20 k of its 52 k clocks outside the tap loop):
//   * one block = a 16 x 16 output patch x BN = 128 channels, 4 waves as 2 (pixel halves) x 2 (channel halves): every
//     wave owns 128 pixels x 64 channels = 8 x 4 MFMA tiles, 64 MFMAs per tap instead of 32 -- the per-tap costs (one
//     barrier, LW = 4 weight DMA instructions per wave, the fragment set-up) are paid once per 64 MFMAs, the per-block
//     costs (set-up, GroupNorm fold, epilogue constants) once per 256 pixels; the 18 x 18 halo is 1.27 x the patch
//     instead of 1.41 x;
//   * 12 fragment reads per 32 MFMAs instead of 16 (8 pixel + 4 channel fragments per k-step feed 32 MFMAs);
//   * the HALO travels global -> LDS by DMA as well (`buffer_load_dwordx4 ... lds`, 41 pieces of 8 pixels x 128 B per
//     channel chunk, XOR swizzle applied on the source side): no staging registers -- which is what makes room for the
//     128 accumulator registers at two waves per SIMD -- and the fused GroupNorm(+act) becomes an in-place LDS pass
//     (read chunk, normalise, write back) between two barriers; padding pixels arrive as zeros (out-of-range buffer
//     offsets) and are skipped by that pass, as the conv pads the ACTIVATED tensor;
//   * fragment reads are software-pipelined in four steps of 16 MFMAs per tap (two k-steps x two pixel groups): the reads
//     of step i + 1 are in flight under the MFMAs of step i, with counted lgkmcnt waits; every address is one of three
//     base registers (+ one XOR for the second k-step) plus an immediate, because the swizzle depends on the halo
//     COLUMN only (chunk ^ (hx & 7)), not on the linear pixel index.
// LDS: halo 41 x 1 KiB + 2 weight slots x 16 KiB + 256 B group statistics = 75 008 B -> two blocks per CU.
#include <atomic>
#include "conv3x3_common.hpp"

namespace {

constexpr int H16_T = 16, H16_W = H16_T + 2, H16_PIX = H16_W * H16_W;   // 18 x 18 = 324 halo pixels

#ifdef H16_STAMPS   // tools/exp/stamps_h16.py: shader-clock stamps of one block's wave 0 (debug build only)
// two clocks per stamp: [i] = s_memtime (shader-clock counter: UNDER-counts while the chip's MFMA pipes are loaded, DESIGN.md
// section 8) and [4096 + i] = s_memrealtime (constant 100 MHz: coarse, but real time); [8192 + 4 b ..] = per-block records of ALL
// blocks: realtime at entry / at the end of the epilogue, HW_REG_HW_ID, HW_REG_LDS_ALLOC (tools/exp/stamps_h16.py --blocks)
__device__ unsigned long long g_h16_stamps[8192 + 4 * 8192];
#define H16_STAMP(i_)                                                                             \
    if (stamp_on) { __builtin_amdgcn_sched_barrier(0); g_h16_stamps[(i_)] = __builtin_readcyclecounter();                  \
                    g_h16_stamps[4096 + (i_)] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define H16_STAMP(i_)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
template <int N> __device__ __forceinline__ void h16_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void h16_wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#endif

// GroupNorm affine + activation of one 16-byte chunk.  f16: the affine runs as packed f16 FMAs on the stored pairs and the
// SiLU as f16 exp / rcp (4 instructions per element instead of ~11; the result is rounded to f16 storage either way).
template <typename T, int EPC>
__device__ __forceinline__ u32x4 h16_gn_act(u32x4 raw, const float* sc, const float* sh, int act) {
    return gn_act_chunk<T, EPC>(raw, sc, sh, act);
}
template <>
__device__ __forceinline__ u32x4 h16_gn_act<f16_t, 8>(u32x4 raw, const float* sc, const float* sh, int act) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    f16x8 x = __builtin_bit_cast(f16x8, raw), o;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const h2 xv = {x[j], x[j + 1]};
        const h2 s = {(_Float16)sc[j], (_Float16)sc[j + 1]}, t = {(_Float16)sh[j], (_Float16)sh[j + 1]};
        h2 z = xv * s + t;
        if (act == 1) {   // z * 1 / (1 + 2^(-z log2 e)): v_pk_mul_f16, 2 v_exp_f16, v_pk_add_f16, 2 v_rcp_f16, v_pk_mul_f16
            const h2 m = z * (h2){(_Float16)-1.442695f, (_Float16)-1.442695f};
            const h2 e = {(_Float16)__builtin_exp2f16(m[0]), (_Float16)__builtin_exp2f16(m[1])};
            const h2 d = e + (h2){(_Float16)1.0f, (_Float16)1.0f};
            z = z * (h2){(_Float16)__builtin_amdgcn_rcph(d[0]), (_Float16)__builtin_amdgcn_rcph(d[1])};
        } else if (act == 2) {
            z[0] = z[0] > (_Float16)0 ? z[0] : (_Float16)0;
            z[1] = z[1] > (_Float16)0 ? z[1] : (_Float16)0;
        }
        o[j] = z[0]; o[j + 1] = z[1];
    }
    return __builtin_bit_cast(u32x4, o);
}

// patch epilogue for MI patch rows per wave (the 8 x 16 kernels use halo_tile_epilogue with MI = 4)
template <typename T, int BN, int MI>
__device__ __forceinline__ void h16_epilogue(const IgemmP& p, f32x4 (&acc)[MI][BN / 32], int b, int py0, int px0, int n0, int z,
                                             float* red) {
    constexpr int NI = BN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const bool want_stats = p.stats != nullptr && p.splitk == 1;
    const int nb = n0 + wn * (BN / 2) + fg * 4;
    f32x4 cs[NI], cq[NI], add[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { cs[j] = f32x4{0.f, 0.f, 0.f, 0.f}; cq[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    epilogue_consts<NI>(p, nb, b, add);
    int mrow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int oy = py0 + wm * MI + i, ox = px0 + frow;
        mrow[i] = (oy < p.OH && ox < p.OW) ? (b * p.OH + oy) * p.OW + ox : -1;
    }
    {
        // (split-K launches take h16_epilogue_splitk.)  Two patch rows at a time: epilogue_tile keeps the residual pieces of ALL its
        // rows in flight, which for MI = 8 x NI = 4 is 128 registers beside the 128 accumulators -- the compiler spilled 600 .. 750
        // VGPRs to scratch around it (every f32-mode launch of this kernel takes this epilogue).  Same operations per element,
        // same order: bit-identical.
#pragma unroll
        for (int h = 0; h < MI; h += 2) {
            const int mr2[2] = {mrow[h], mrow[h + 1]};
            epilogue_tile<T, 2, NI>(p, mr2, nb, add, false, *reinterpret_cast<f32x4 (*)[2][NI]>(&acc[h]));
        }
        if (want_stats) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (mrow[i] < 0) continue;
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if (nb + 16 * j < p.N) { cs[j] += acc[i][j]; cq[j] += acc[i][j] * acc[i][j]; }
            }
        }
    }
    if (want_stats) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cs[j][r] = row16_sum(cs[j][r]);
                cq[j][r] = row16_sum(cq[j][r]);
            }
            if (frow == 0) {
                float* dst = red + ((wm * BN) + wn * (BN / 2) + j * 16 + fg * 4) * 2;
#pragma unroll
                for (int r = 0; r < 4; ++r) { dst[2 * r] = cs[j][r]; dst[2 * r + 1] = cq[j][r]; }
            }
        }
        __syncthreads();
        for (int c = tid; c < 2 * BN; c += 256) {
            const int n = n0 + (c >> 1);
            if (n < p.N) atomicAdd(p.stats + ((size_t)b * p.N + n0) * 2 + c, (double)red[c] + (double)red[2 * BN + c]);
        }
    }
}

// split-K slices: the f32 partial tile goes to the workspace slab of slice z, nothing else (bias, residual, statistics belong
// to the reduction).  A kernel instantiation of its own (SK): inside h16_epilogue the store loop shares its registers with the
// full epilogue's live ranges and the compiler spilled 600 .. 900 VGPRs around it (467 scratch instructions per wave).
template <int BN, int MI>
__device__ __forceinline__ void h16_epilogue_splitk(const IgemmP& p, const f32x4 (&acc)[MI][BN / 32], int b, int py0, int px0, int n0,
                                                    int z) {
    constexpr int NI = BN / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const int nb = n0 + wn * (BN / 2) + fg * 4;
    const int ox = px0 + frow;
    float* const slab = p.ws + (size_t)z * p.M * p.N + nb;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int oy = py0 + wm * MI + i;
        if (oy < p.OH && ox < p.OW) {
            float* row = slab + (size_t)((b * p.OH + oy) * p.OW + ox) * p.N;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const f32x4 v = acc[i][j];
                if (nb + 16 * j < p.N) *reinterpret_cast<float4*>(row + 16 * j) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// The same for 16-bit outputs without split-K / GEGLU: the finished values (bias, time row, residual, ReLU applied in
// registers as above) are rounded to T, written to an LDS tile [256 pixels][BN channels] (rows padded by 16 B) and leave as
// 16-byte stores, 16 lanes per pixel row = full 256-byte lines -- the direct path stores 8-byte pieces of 16 different rows
// per instruction and spent ~25 k clocks per block (in-kernel stamps), a quarter of a K = 1152 layer's block time.
typedef __attribute__((address_space(3))) char h16_lds_char;

template <typename T, int BN, int MI>
__device__ __forceinline__ void h16_epilogue_lds(const IgemmP& p, f32x4 (&acc)[MI][BN / 32], int b, int py0, int px0, int n0,
                                                 char* lds, h16_lds_char* lds3) {
    constexpr int NI = BN / 32;
    constexpr int ROW = BN * (int)sizeof(T) + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const bool want_stats = p.stats != nullptr;
    const int nb = n0 + wn * (BN / 2) + fg * 4;
    float* red = reinterpret_cast<float*>(lds + 256 * ROW);
#ifdef H16_STAMPS
    const bool stamp_on = blockIdx.x == gridDim.x / 2 + 1 && blockIdx.z == 0 && tid == 0;
#endif
    H16_STAMP(2000);
    // Every optional piece sits behind a block-uniform branch around its own loop: written as per-element conditions the
    // compiler turned them into selects (v_max + v_cndmask per value for the optional ReLU, masks on every statistic) and
    // the epilogue issued ~2 800 VALU instructions, 16 k clocks next to a co-resident block (in-kernel stamps).
    const bool full = __builtin_amdgcn_readfirstlane((py0 + H16_T <= p.OH && px0 + H16_T <= p.OW && n0 + BN <= p.N) ? 1 : 0) != 0;
    f32x4 cs[NI], cq[NI], add[NI];
    epilogue_consts<NI>(p, nb, b, add);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] += add[j];
    if (p.residual) {
#if defined(__HIP_DEVICE_COMPILE__)
        // The residual tile [256 pixels][BN channels] comes in by LDS-DMA (full 256-byte lines, 64 one-KiB pieces = 4 pixel
        // rows each, 16 per wave) and the lanes pick their 8-byte pieces out of the LDS: as 32 eight-byte global gathers per
        // lane it cost 13 k clocks per block (stamps; the gathers of one wave cover 16 KB against a 32 KB L1).  The 16-byte
        // chunk c of pixel row px lands at position c ^ (px & 15) (source-side swizzle, like the halo): the 16 pixel columns
        // a read instruction covers hit 16 different positions.
        static_assert(BN * sizeof(T) == 256, "residual DMA: one pixel row = 16 chunks");
        const size_t rbytes = (size_t)p.M * (size_t)p.ldr * sizeof(T);
        const __amdgpu_buffer_rsrc_t rsr =
            __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, (unsigned)(rbytes < 0xffffffffull ? rbytes : 0xffffffffull),
                                              0x00020000);
        const int pl = lane >> 4, pos = lane & 15;                 // pixel of the piece, LDS chunk position
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int oy = py0 + wave * 4 + (t >> 2), ox = px0 + 4 * (t & 3) + pl;
            const int c = pos ^ ((4 * (t & 3) + pl) & 15);        // global chunk that belongs at this position
            const int n = n0 + c * (16 / (int)sizeof(T));
            unsigned off = (unsigned)((((size_t)(b * p.OH + oy) * p.OW + ox) * p.ldr + n) * sizeof(T));
            if (!full && (oy >= p.OH || ox >= p.OW || n >= p.N)) off = 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsr, lds3 + (wave * 16 + t) * 1024, 16, off, 0, 0, 0);
        }
        h16_wait_vmcnt<0>();
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int px = (wm * MI + i) * 16 + frow;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int c = wn * 8 + j * 2 + (fg >> 1);
                acc[i][j] += load4<T>(reinterpret_cast<const T*>(lds + px * 256 + ((c ^ frow) * 16) + (fg & 1) * 8));
            }
        }
        __syncthreads();   // the tile region is rewritten with the outputs below
#endif
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
        char* row = lds + ((wm * MI + i) * 16 + frow) * ROW + (wn * (BN / 2) + fg * 4) * (int)sizeof(T);
#pragma unroll
        for (int j = 0; j < NI; ++j) store4<T>(reinterpret_cast<T*>(row + j * 16 * (int)sizeof(T)), acc[i][j]);
    }
    H16_STAMP(2001);
    if (want_stats) {
#pragma unroll
        for (int j = 0; j < NI; ++j) { cs[j] = f32x4{0.f, 0.f, 0.f, 0.f}; cq[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (full) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) { cs[j] += acc[i][j]; cq[j] += acc[i][j] * acc[i][j]; }
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const bool inside = py0 + wm * MI + i < p.OH && px0 + frow < p.OW;
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if (inside && nb + 16 * j < p.N) { cs[j] += acc[i][j]; cq[j] += acc[i][j] * acc[i][j]; }
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cs[j][r] = row16_sum(cs[j][r]);
                cq[j][r] = row16_sum(cq[j][r]);
            }
            if (frow == 0) {
                float* dst = red + ((wm * BN) + wn * (BN / 2) + j * 16 + fg * 4) * 2;
#pragma unroll
                for (int r = 0; r < 4; ++r) { dst[2 * r] = cs[j][r]; dst[2 * r + 1] = cq[j][r]; }
            }
        }
    }
    H16_STAMP(2002);
    __syncthreads();
    H16_STAMP(2003);
    constexpr int CPR = BN * (int)sizeof(T) / 16;             // 16-byte chunks per pixel row of the tile
#pragma unroll 4
    for (int c = tid; c < 256 * CPR; c += 256) {
        const int px = c / CPR, ch = c - px * CPR;
        const int oy = py0 + (px >> 4), ox = px0 + (px & 15);
        const int n = n0 + ch * (16 / (int)sizeof(T));
        if (oy < p.OH && ox < p.OW && n < p.N) {
            const uint4 v = *reinterpret_cast<const uint4*>(lds + px * ROW + ch * 16);
            *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + (size_t)((b * p.OH + oy) * p.OW + ox) * p.ldo + n) = v;
        }
    }
    H16_STAMP(2004);
    if (want_stats) {
        for (int c = tid; c < 2 * BN; c += 256) {
            const int n = n0 + (c >> 1);
            if (n < p.N) atomicAdd(p.stats + ((size_t)b * p.N + n0) * 2 + c, (double)red[c] + (double)red[2 * BN + c]);
        }
    }
    H16_STAMP(2005);
}

template <typename T, int BN, bool FUSE, bool WIDE, bool SK>
__global__ __launch_bounds__(256, 2) void conv3x3_h16_kernel(const IgemmP p, const PatchDecode pd) {
    kernarg_touch<5>();
#if defined(__HIP_DEVICE_COMPILE__)
    // (Round 6, measured and removed -- tools/exp/r6_h16_stagger_prio.patch, profiles/round6_h16_realtime.txt: delaying the
    // workgroup in the CU's upper LDS allocation by 1.5 .. 16 k clocks, and running its tap loop at s_setprio 1 .. 3, both
    // change no layer by more than the +-2 % run-to-run spread: the two resident workgroups are not in a harmful lockstep.)
    constexpr int EPC = TT<T>::EPC;
    constexpr int BKE = 8 * EPC;
    constexpr int MI = 8, NI = BN / 32;
    constexpr int LW = BN / 32;                              // weight DMA instructions per wave and tap
    constexpr int NPIECE = (H16_PIX + 7) / 8;                // 41 halo pieces of 8 pixels x 128 B
    constexpr int HP = (NPIECE + 3) / 4;                     // per wave (11)
    constexpr int HALO_B = NPIECE * 1024, W_B = BN * 128;
    constexpr int TP = (H16_PIX + 31) / 32;                  // transform iterations per thread (32 pixels x 8 chunks a round)
    constexpr int ROWB = H16_W * 128;                        // 2304 bytes per halo row (low 8 bits zero: commutes with ^ 64)
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem_raw;
    const unsigned lds_base = (unsigned)(size_t)lds0;
    const unsigned wring = lds_base + HALO_B;
    const unsigned gstat_addr = wring + 2 * W_B;
    float2* gstat = reinterpret_cast<float2*>(smem_raw + HALO_B + 2 * W_B);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const int z = blockIdx.z;

    int bid = blockIdx.x;   // XCD-aware order (see igemm.hip)
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
#ifdef H16_STAMPS
    const bool stamp_on = blockIdx.x == gridDim.x / 2 + 1 && blockIdx.z == 0 && tid == 0;
    int stamp_i = 8;
#endif
    H16_STAMP(0);
#ifdef H16_STAMPS
    if (stamp_on) g_h16_stamps[7] = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    if (tid == 0 && blockIdx.z == 0 && blockIdx.x < 1024) g_h16_stamps[3000 + blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    if (tid == 0 && blockIdx.z == 0 && blockIdx.x < 8192) {
        g_h16_stamps[8192 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        g_h16_stamps[8192 + 4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        g_h16_stamps[8192 + 4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    }
#endif
    const int tm = magic_div(bid, pd.m_tilesN), tn = bid - tm * p.tilesN;
    const int n0 = tn * BN;
    const int b = magic_div(tm, pd.m_ppi);
    const int pr = tm - b * pd.patchesPerImg;
    const int pry = magic_div(pr, pd.m_px);
    const int py0 = pry * H16_T, px0 = (pr - pry * pd.patchesX) * H16_T;

    // ---- halo DMA state: piece q = wave + 4 i; lane l lands LDS slot (pixel 8 q + (l >> 3), 16-byte slot l & 7), which
    //      must hold global chunk (l & 7) ^ (hx & 7) of that pixel.  pk = pixel offset * 8 + global chunk, -1 = zeros ----
    //      (recomputed per piece: ~12 VALU each, 11 pieces per wave and chunk -- registers are the scarce resource here)
    const int hl = lane >> 3, cl = lane & 7;
    // ---- weight DMA state: instruction i of this wave lands rows 8 * (wave * LW + i) .. + 7 of the tile; row r of them at
    //      voffset base + i * 8 rows (the swizzle (row & 7) = lane >> 3 is the same for every i) ----
    const int wrow0 = wave * LW * 8 + hl;
    const unsigned wv0 = (unsigned)(((size_t)(n0 + wrow0) * p.ldw + (cl ^ hl) * EPC) * sizeof(T));
    const unsigned wvstep = (unsigned)__builtin_amdgcn_readfirstlane((int)((size_t)8 * p.ldw * sizeof(T)));
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in2 ? p.in2 : p.in1), 0,
                                                                         p.in2 ? p.bytes2 : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.bytesw, 0x00020000);

    const int ush = __builtin_amdgcn_readfirstlane(p.upsample ? 1 : 0);   // (never together with FUSE: the launcher checks)
    const int nchunks = p.Ctot / BKE;
    const int ck0 = (nchunks * z) / p.splitk;
    const int ck1 = (nchunks * (z + 1)) / p.splitk;

#define H16_DMA_HALO(ck)                                                                                \
    {                                                                                                   \
        const int c0_ = (ck) * BKE;                                                                     \
        const bool first_ = c0_ < p.C1;                                                                 \
        const __amdgpu_buffer_rsrc_t rs_ = first_ ? rs1 : rs2;                                          \
        const int ld_ = first_ ? p.ld1 : p.ld2;                                                         \
        const int cofs_ = first_ ? c0_ : c0_ - p.C1;                                                    \
        int hlv_ = hl;   /* opaque copy: keeps the per-piece address arithmetic inside the loop (hoisted, it is 20+ */ \
        asm volatile("" : "+v"(hlv_));   /* registers that spill out of the MFMA loop) */               \
        _Pragma("unroll") for (int i = 0; i < HP; ++i) {                                                \
            if (wave + 4 * i < NPIECE) {                                                                \
                const int h_ = (wave + 4 * i) * 8 + hlv_;                                               \
                const int hy_ = h_ / H16_W, hx_ = h_ - hy_ * H16_W;                                     \
                const int iy_ = py0 - 1 + hy_, ix_ = px0 - 1 + hx_;                                     \
                /* nearest-2x upsample folded into the gather (ush = 1): halo pixel (iy, ix) of the upsampled map reads    */ \
                /* source pixel (iy >> 1, ix >> 1); the zero padding lies outside the UPSAMPLED map                     */ \
                const bool ok_ = h_ < H16_PIX && (unsigned)iy_ < (unsigned)(p.IH << ush) &&                 \
                                 (unsigned)ix_ < (unsigned)(p.IW << ush);                                   \
                const unsigned off_ = (unsigned)(((b * p.IH + (iy_ >> ush)) * p.IW + (ix_ >> ush)) * ld_ + cofs_ + \
                                                 (cl ^ (hx_ & 7)) * EPC) * (unsigned)sizeof(T);             \
                lds_char* dst_ = lds0 + (wave + 4 * i) * 1024;                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, dst_, 16, ok_ ? off_ : OOB, 0, 0, 0);     \
            }                                                                                           \
        }                                                                                               \
    }
    // GroupNorm(+act) of the landed chunk, in place: thread = (pixel (tid >> 3) + 32 i, global chunk tid & 7)
#define H16_TRANSFORM_LOOP(ACT)                                                                         \
        _Pragma("unroll") for (int i = 0; i < TP; ++i) {                                                \
            const int h_ = tp_ + 32 * i;                                                                \
            const int hy_ = h_ / H16_W, hx_ = h_ - hy_ * H16_W;                                         \
            const int iy_ = py0 - 1 + hy_, ix_ = px0 - 1 + hx_;                                         \
            if (h_ < H16_PIX && (unsigned)iy_ < (unsigned)p.IH && (unsigned)ix_ < (unsigned)p.IW) {     \
                uint4* q_ = reinterpret_cast<uint4*>(smem_raw + h_ * 128 + ((cpos_ ^ (hx_ & 7)) * 16));  \
                const u32x4 v_ = __builtin_bit_cast(u32x4, *q_);                                        \
                *q_ = __builtin_bit_cast(uint4, h16_gn_act<T, EPC>(v_, sc_, sh_, ACT));                 \
            }                                                                                           \
        }
#define H16_TRANSFORM(ck)                                                                               \
    {                                                                                                   \
        const int cpos_ = tid & 7;                                                                      \
        const unsigned cb_ = (unsigned)((ck) * BKE + cpos_ * EPC);                                      \
        float sc_[EPC], sh_[EPC];                                                                       \
        {                                                                                               \
            const float* gs_ = p.gn_gamma + cb_;                                                        \
            const float* gh_ = p.gn_beta + cb_;                                                         \
            float2 st_[EPC];                                                                            \
            _Pragma("unroll") for (int j = 0; j < EPC; ++j) {                                           \
                const unsigned ga_ = gstat_addr + __umulhi(cb_ + j, p.gn_magic) * 8u;                   \
                asm volatile("ds_read_b64 %0, %1" : "=v"(st_[j]) : "v"(ga_));                           \
            }                                                                                           \
            _Pragma("unroll") for (int j = 0; j < EPC; j += 4) {                                        \
                const float4 a_ = *reinterpret_cast<const float4*>(gs_ + j);                            \
                const float4 b_ = *reinterpret_cast<const float4*>(gh_ + j);                            \
                sc_[j] = a_.x; sc_[j + 1] = a_.y; sc_[j + 2] = a_.z; sc_[j + 3] = a_.w;                 \
                sh_[j] = b_.x; sh_[j + 1] = b_.y; sh_[j + 2] = b_.z; sh_[j + 3] = b_.w;                 \
            }                                                                                           \
            h16_wait_lgkmcnt<0>();                                                                      \
            _Pragma("unroll") for (int j = 0; j < EPC; ++j) {                                           \
                const float s_ = st_[j].y * sc_[j];                                                     \
                sh_[j] = sh_[j] - st_[j].x * s_;                                                        \
                sc_[j] = s_;                                                                            \
            }                                                                                           \
        }                                                                                               \
        H16_STAMP(6);                                                                                   \
        int tp_ = tid >> 3;                                                                             \
        asm volatile("" : "+v"(tp_));                                                                   \
        /* plain LDS accesses: no DMA is in flight during this pass, so the compiler may schedule them freely (the pass is */ \
        /* VALU-bound: ~11 chunks x ~90 instructions per thread; batching the LDS round trips changed nothing)            */ \
        /* the activation switch sits OUTSIDE the pass (three copies of the loop): tested per pair inside it, it cost 120   */ \
        /* scalar branches + 264 s_nop per thread and chunk                                                               */ \
        if (p.act == 1) { H16_TRANSFORM_LOOP(1) } else if (p.act == 2) { H16_TRANSFORM_LOOP(2) } else { H16_TRANSFORM_LOOP(0) } \
    }
    // weight stream: tile s = (chunk lck, tap ltap) goes to ring slot s & 1
    int lck = ck0, ltap = 0, lslot = 0;
#define H16_DMA_W()                                                                                     \
    {                                                                                                   \
        const unsigned kofs_ = (unsigned)__builtin_amdgcn_readfirstlane((ltap * p.Ctot + lck * BKE) * (int)sizeof(T)); \
        _Pragma("unroll") for (int i = 0; i < LW; ++i) {                                                \
            lds_char* dst_ = lds0 + HALO_B + lslot * W_B + (wave * LW + i) * 1024;                      \
            /* rows n >= N lie beyond the weight buffer's byte count: the buffer bounds check lands zeros, no select and */ \
            /* no per-piece offset registers (spilled, they serialised the four DMA issues behind scratch reloads) */   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst_, 16, wv0, kofs_ + (unsigned)i * wvstep, 0, 0); \
        }                                                                                               \
        if (++ltap == 9) { ltap = 0; ++lck; }                                                           \
        lslot ^= 1;                                                                                     \
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment base addresses: pixel column frow + sx of patch row wm * 8, chunk fg (k-step 1: ^ 64)
    unsigned ab[3];
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
        const int hx_ = frow + sx;
        ab[sx] = lds_base + (unsigned)(((wm * MI) * H16_W + hx_) * 128 + ((fg ^ (hx_ & 7)) * 16));
    }
    const unsigned wf0 = wring + (unsigned)(((wn * (BN / 2) + frow) * 8 + (fg ^ (frow & 7))) * 16);

    H16_STAMP(1);
    if (ck1 > ck0) {
        H16_DMA_HALO(ck0);
        H16_DMA_W();
        H16_STAMP(2);
        if (FUSE) gn_fold_groups(p, b, gstat);
    }
    H16_STAMP(3);
    int wslot = 0;   // ring slot of the tap about to run
    for (int ck = ck0; ck < ck1; ++ck) {
        // ---- the chunk's halo has been requested: land it, normalise it ----
        H16_STAMP(stamp_i); 
        h16_wait_vmcnt<0>();
        __syncthreads();
        H16_STAMP(stamp_i + 1);
        if (FUSE) {
            // the pass is pure VALU + LDS on the block's critical path while the CU's other block sits in its MFMA loop:
            // take the issue arbitration (priority, then age -- MI355X_MICROARCH.md "Two waves per SIMD") for its duration
            __builtin_amdgcn_s_setprio(3);
            H16_TRANSFORM(ck);
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
        }
        H16_STAMP(stamp_i + 2);
#ifdef H16_STAMPS
        stamp_i += 3;
#endif
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bool more = (tap < 8) || (ck + 1 < ck1);
            if (tap > 0) {                    // tap 0's tile was waited for together with the halo
                h16_wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();  // this tap's tile visible; the other slot (read by tap - 1) is free
            }
            H16_STAMP(stamp_i);
            if (more) H16_DMA_W();
            __builtin_amdgcn_sched_barrier(0);
            H16_STAMP(stamp_i + 1);
            {
                const int r_ = tap / 3, sx_ = tap % 3;
                const unsigned a0 = ab[sx_], a1 = ab[sx_] ^ 64u;
                const unsigned w0 = wf0 + (unsigned)wslot * (unsigned)W_B, w1 = w0 ^ 64u;
                u32x4 af[2][4], wf[2][NI];
                // step 0 operands + step 1 pixels
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[0][i]) : "v"(a0), "n"((r_ + i) * ROWB));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[0][j]) : "v"(w0), "n"(j * 2048));
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[1][i]) : "v"(a0), "n"((r_ + 4 + i) * ROWB));
                h16_wait_lgkmcnt<4>();
                __builtin_amdgcn_sched_barrier(0);
                H16_STAMP(stamp_i + 2);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        mma16<T>(__builtin_bit_cast(uint4, wf[0][j]), __builtin_bit_cast(uint4, af[0][i]), acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                // step 2 operands under step 0 / 1 MFMAs
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[0][i]) : "v"(a1), "n"((r_ + i) * ROWB));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[1][j]) : "v"(w1), "n"(j * 2048));
                h16_wait_lgkmcnt<4 + NI>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        mma16<T>(__builtin_bit_cast(uint4, wf[0][j]), __builtin_bit_cast(uint4, af[1][i]), acc[4 + i][j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[1][i]) : "v"(a1), "n"((r_ + 4 + i) * ROWB));
                h16_wait_lgkmcnt<4>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        mma16<T>(__builtin_bit_cast(uint4, wf[1][j]), __builtin_bit_cast(uint4, af[0][i]), acc[i][j]);
                h16_wait_lgkmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        mma16<T>(__builtin_bit_cast(uint4, wf[1][j]), __builtin_bit_cast(uint4, af[1][i]), acc[4 + i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            H16_STAMP(stamp_i + 3);
#ifdef H16_STAMPS
            stamp_i += 4;
#endif
            wslot ^= 1;
        }
        if (ck + 1 < ck1) {
            __builtin_amdgcn_s_barrier();     // single halo buffer: every wave has read its last fragments of this chunk
            H16_DMA_HALO(ck + 1);
        }
    }
#undef H16_DMA_HALO
#undef H16_TRANSFORM
#undef H16_DMA_W
    H16_STAMP(4);
    h16_wait_vmcnt<0>();
    __syncthreads();   // the LDS becomes the statistics scratch of the epilogue
    // WIDE (chosen by the launcher): 16-bit outputs without split-K / GEGLU whose rows keep 16-byte alignment
    // the epilogue is VALU work on the block's critical path like the GroupNorm pass: raised issue priority beside the partner
    // workgroup's MFMA stream (same box, two runs each: 361.0 / 361.8 -> 363.1 / 362.9 images/s; priority 1: 360.3 / 361.1)
    __builtin_amdgcn_s_setprio(3);
    if constexpr (SK) h16_epilogue_splitk<BN, MI>(p, acc, b, py0, px0, n0, z);
    else if constexpr (WIDE) h16_epilogue_lds<T, BN, MI>(p, acc, b, py0, px0, n0, smem_raw, lds0);
    else h16_epilogue<T, BN, MI>(p, acc, b, py0, px0, n0, z, reinterpret_cast<float*>(smem_raw));
    H16_STAMP(5);
#ifdef H16_STAMPS
    if (tid == 0 && blockIdx.z == 0 && blockIdx.x < 8192) g_h16_stamps[8192 + 4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
#endif
}

}  // namespace

#if defined(H16_STAMPS) && !defined(H16_ONLY_F32)
extern "C" int madm_debug_read_h16_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_h16_stamps), sizeof(unsigned long long) * n);
}
#endif

namespace {

template <typename T, int BN, bool FUSE, bool WIDE, bool SK = false>
int launch_h16_one(const IgemmP& p0, hipStream_t s) {
    IgemmP p = p0;
    constexpr size_t lds0 = (size_t)((H16_PIX + 7) / 8) * 1024 + 2 * (size_t)BN * 128 + 32 * sizeof(float2);
    // experiment (MADM_EXP_H16_LDS=<bytes>): ask for more LDS than the kernel uses, e.g. 90112 -> ONE workgroup per CU, the
    // rest of the CU stays free for the workgroups of kernels on other streams (staged pipeline)
    static const size_t lds_exp = [] { const char* e = getenv("MADM_EXP_H16_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();
    const size_t lds = lds_exp > lds0 ? lds_exp : lds0;
    auto kern = conv3x3_h16_kernel<T, BN, FUSE, WIDE, SK>;
    // the attribute is per device: one bit per device id, set once (atomic: host threads may launch concurrently)
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(lds > 64 * 1024 ? lds : 64 * 1024));
        if (e != hipSuccess) {
            madm_set_error("conv3x3 (16 x 16 patches): cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
            return MADM_ERR_LAUNCH;
        }
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    const int patchesX = (p.OW + H16_T - 1) / H16_T, patchesY = (p.OH + H16_T - 1) / H16_T;
    p.tilesN = (p.N + BN - 1) / BN;
    dim3 grid((unsigned)(p.B * patchesX * patchesY * p.tilesN), 1, (unsigned)p.splitk);
    PatchDecode pd;
    if (!patch_decode_fill(pd, patchesX, patchesY, p.tilesN, (long long)grid.x)) {
        madm_set_error("conv3x3 (16 x 16 patches): grid of %u blocks too large for the reciprocal patch decode", grid.x);
        return MADM_ERR_INVALID_ARG;
    }
    kern<<<grid, 256, lds, s>>>(p, pd);
    return madm_check_launch("conv3x3_h16_kernel");
}

}  // namespace

template <typename T>
int launch_conv3x3_h16(const IgemmP& p, int bn, hipStream_t s) {
    const bool fuse = p.gn_sums1 != nullptr;
    if (fuse && p.upsample) {
        madm_set_error("conv3x3 (16 x 16 patches): GroupNorm fusion and the upsample gather do not combine");
        return MADM_ERR_UNSUPPORTED;
    }
    (void)bn;
    if (p.splitk > 1) return fuse ? launch_h16_one<T, 128, true, false, true>(p, s) : launch_h16_one<T, 128, false, false, true>(p, s);
    if constexpr (sizeof(T) == 2) {
        if (p.splitk == 1 && !p.out_f32 && p.epilogue != MADM_EPI_GEGLU && (p.ldo & 7) == 0 && (p.N & 7) == 0)
            return fuse ? launch_h16_one<T, 128, true, true>(p, s) : launch_h16_one<T, 128, false, true>(p, s);
    }
    return fuse ? launch_h16_one<T, 128, true, false>(p, s) : launch_h16_one<T, 128, false, false>(p, s);
}
// The f32 instantiation lives in a translation unit of its own (conv3x3_h16_f32.hip includes this file with H16_ONLY_F32),
// compiled WITHOUT packed-FP32 VALU ops like every other f32-mode kernel of the library (Makefile): the f32
// arithmetic is the parity mode, and round 5 saw one unexplained f32 mismatch of an eval forward (first process on a fresh box)
// that neither the LDS nor the HBM poison harness reproduces -- the packed ops stay only where they pay, in the 16-bit
// instantiations, whose forms tools/isa_pk_scan.py checks.
#ifdef H16_ONLY_F32
template int launch_conv3x3_h16<float>(const IgemmP&, int, hipStream_t);
#else
template int launch_conv3x3_h16<bf16_t>(const IgemmP&, int, hipStream_t);
template int launch_conv3x3_h16<f16_t>(const IgemmP&, int, hipStream_t);
#endif
