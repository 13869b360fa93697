// LDS-DMA variant of the halo-tile 3x3 convolution (conv3x3.hip): same patch geometry, fused GroupNorm and epilogue,
// but the WEIGHT tiles travel global -> LDS directly (`buffer_load_dwordx4 ... lds`, 8 rows x 128 B per wave
// instruction) into a three-slot ring two taps ahead of the MFMAs: no staging registers and no ds_write_b128 phase
// (~390 of the ~1650 clocks of a tap step in the register-staged kernel, tools/exp/stamps.py).  The LDS the third
// weight slot needs comes from single-buffering the halo: the next channel chunk is written after one extra barrier
// at tap 8 (its transform already sat on the critical path there).  2 blocks per CU as before:
//   LDS = halo 180 px x 128 B (23 040) + 3 x BN x 128 B + 256 B group statistics = 72 448 B (BN 128) / 47 872 B (BN 64).
// Every LDS access that runs while DMA is in flight is inline asm: the compiler cannot tell the ring slots from the
// halo and would drain vmcnt(0) in front of any ds_read / ds_write it can see.  The weight tile of tap s is waited
// for with a counted vmcnt (only the next tile's loads may stay outstanding) right before the tap's one barrier.
#include "conv3x3_common.hpp"

namespace {

#if defined(__HIP_DEVICE_COMPILE__)
template <int N> __device__ __forceinline__ void c3_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void c3_wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#endif

#ifdef C3D_STAMPS   // tools/exp/stamps_c3d.py: block timeline (block 5 / thread 0) + per-tap stamps of the first chunks
__device__ unsigned long long g_c3d_stamps[256];
#define C3D_STAMP(k_) if (blockIdx.x == 5 && blockIdx.z == 0 && threadIdx.x == 0 && (k_) < 256) g_c3d_stamps[(k_)] = __builtin_readcyclecounter();
#else
#define C3D_STAMP(k_)
#endif

template <typename T, int BN, bool FUSE>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_dma_kernel(const IgemmP p, const PatchDecode pd) {
    C3D_STAMP(0);
    kernarg_touch<5>();
#if defined(__HIP_DEVICE_COMPILE__)   // LDS address-space casts and gfx asm: device pass only (the host needs the stub)
    constexpr int EPC = TT<T>::EPC;
    constexpr int BKE = 8 * EPC;
    constexpr int MI = 4, NI = BN / 32;     // wave tile: 4 patch rows (64 px) x BN/2 channels
    constexpr int LW = BN / 32;             // weight wave-instructions (8 rows each) per wave and tap
    constexpr int HI = (HPIX * 8 + 255) / 256;  // halo chunks staged per thread (6)
    constexpr int HALO_U4 = HPIX * 8, W_U4 = BN * 8;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem_raw;
    const unsigned lds_base = (unsigned)(size_t)lds0;
    const unsigned wring = lds_base + HALO_U4 * 16;                      // byte address of weight slot 0
    const unsigned gstat_addr = wring + 3 * W_U4 * 16;                   // float2[32] {mean, rstd}
    float2* gstat = reinterpret_cast<float2*>(smem_raw + (HALO_U4 + 3 * W_U4) * 16);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int cpos = tid & 7;
    const int frow = lane & 15, fg = lane >> 4;
    const int z = blockIdx.z;

    int bid = blockIdx.x;   // XCD-aware order (see igemm.hip)
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int tm = magic_div(bid, pd.m_tilesN), tn = bid - tm * p.tilesN;
    const int n0 = tn * BN;
    const int b = magic_div(tm, pd.m_ppi);
    const int pr = tm - b * pd.patchesPerImg;
    const int pry = magic_div(pr, pd.m_px);
    const int py0 = pry * TH, px0 = (pr - pry * pd.patchesX) * TW;

    // ---- halo staging state (as conv3x3.hip) ----
    int pixoff[HI];
    unsigned haddr[HI];   // LDS byte address of the staged chunk, 0xffffffff = beyond the halo
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int idx = tid + 256 * i;
        const int h = idx >> 3;
        const int hy = h / HWD, hx = h - hy * HWD;
        const int iy = py0 - 1 + hy, ix = px0 - 1 + hx;
        const bool ok = idx < HPIX * 8 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        pixoff[i] = ok ? (b * p.IH + iy) * p.IW + ix : -1;
        haddr[i] = idx < HPIX * 8 ? lds_base + (unsigned)(h * 8 + (cpos ^ (h & 7))) * 16u : 0xffffffffu;
    }
    // ---- weight DMA state: instruction i of this wave lands rows 8 * (wave * LW + i) .. + 7 of the tile ----
    unsigned wvoff[LW];
#pragma unroll
    for (int i = 0; i < LW; ++i) {
        const int row = (wave * LW + i) * 8 + (lane >> 3);
        const int n = n0 + row;
        wvoff[i] = (n < p.N) ? (unsigned)(((size_t)n * p.ldw + ((lane & 7) ^ (row & 7)) * EPC) * sizeof(T)) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in2 ? p.in2 : p.in1), 0,
                                                                         p.in2 ? p.bytes2 : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.bytesw, 0x00020000);

    const int nchunks = p.Ctot / BKE;
    const int ck0 = (nchunks * z) / p.splitk;
    const int ck1 = (nchunks * (z + 1)) / p.splitk;
    const int S = (ck1 - ck0) * 9;

    u32x4 hr[HI];
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) { sc[j] = 1.f; sh[j] = 0.f; }

#define D3_LOAD_HALO(ck)                                                                            \
    {                                                                                               \
        const int c0_ = (ck) * BKE;                                                                 \
        const bool first_ = c0_ < p.C1;                                                             \
        const __amdgpu_buffer_rsrc_t rs_ = first_ ? rs1 : rs2;                                      \
        const int ld_ = first_ ? p.ld1 : p.ld2;                                                     \
        const int cofs_ = (first_ ? c0_ : c0_ - p.C1) + cpos * EPC;                                 \
        _Pragma("unroll") for (int i = 0; i < HI; ++i) {                                            \
            const unsigned off_ = (unsigned)(pixoff[i] * ld_ + cofs_) * (unsigned)sizeof(T);        \
            hr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_, pixoff[i] >= 0 ? off_ : OOB, 0, 0);  \
        }                                                                                           \
        if (FUSE) {                                                                                 \
            const float* gs_ = p.gn_gamma + c0_ + cpos * EPC;                                       \
            const float* gh_ = p.gn_beta + c0_ + cpos * EPC;                                        \
            _Pragma("unroll") for (int j = 0; j < EPC; j += 4) {                                    \
                const float4 a_ = *reinterpret_cast<const float4*>(gs_ + j);                        \
                const float4 b_ = *reinterpret_cast<const float4*>(gh_ + j);                        \
                sc[j] = a_.x; sc[j + 1] = a_.y; sc[j + 2] = a_.z; sc[j + 3] = a_.w;                 \
                sh[j] = b_.x; sh[j + 1] = b_.y; sh[j + 2] = b_.z; sh[j + 3] = b_.w;                 \
            }                                                                                       \
        }                                                                                           \
    }
// normalise (FUSE) and write the staged chunk to the (single) halo buffer; all LDS traffic as asm
#define D3_STORE_HALO(ck_)                                                                          \
    {                                                                                               \
        if (FUSE) {                                                                                 \
            const unsigned cb_ = (unsigned)((ck_) * BKE + cpos * EPC);                              \
            float2 st_[EPC];                                                                        \
            _Pragma("unroll") for (int j = 0; j < EPC; ++j) {                                       \
                const unsigned ga_ = gstat_addr + __umulhi(cb_ + j, p.gn_magic) * 8u;               \
                asm volatile("ds_read_b64 %0, %1" : "=v"(st_[j]) : "v"(ga_));                       \
            }                                                                                       \
            c3_wait_lgkmcnt<0>();                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            _Pragma("unroll") for (int j = 0; j < EPC; ++j) {                                       \
                const float s_ = st_[j].y * sc[j];                                                  \
                sh[j] = sh[j] - st_[j].x * s_;                                                      \
                sc[j] = s_;                                                                         \
            }                                                                                       \
        }                                                                                           \
        _Pragma("unroll") for (int i = 0; i < HI; ++i) {                                            \
            if (haddr[i] != 0xffffffffu) {                                                          \
                u32x4 v_ = hr[i];                                                                   \
                if (FUSE) {                                                                         \
                    v_ = gn_act_chunk<T, EPC>(v_, sc, sh, p.act);                                   \
                    if (pixoff[i] < 0) v_ = u32x4{0u, 0u, 0u, 0u};   /* the conv pads the ACTIVATED tensor */ \
                }                                                                                   \
                asm volatile("ds_write_b128 %0, %1" ::"v"(haddr[i]), "v"(v_) : "memory");           \
            }                                                                                       \
        }                                                                                           \
        c3_wait_lgkmcnt<0>();                                                                       \
    }
    // weight stream: (lck, ltap) = tile the next D3_DMA_W fetches; tile s goes to ring slot s % 3
    int lck = ck0, ltap = 0, lslot = 0;
#define D3_DMA_W()                                                                                  \
    {                                                                                               \
        const unsigned kofs_ = (unsigned)(ltap * p.Ctot + lck * BKE) * (unsigned)sizeof(T);         \
        _Pragma("unroll") for (int i = 0; i < LW; ++i) {                                            \
            lds_char* dst_ = lds0 + (HALO_U4 + lslot * W_U4 + (wave * LW + i) * 64) * 16;           \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst_, 16, wvoff[i], kofs_, 0, 0);         \
        }                                                                                           \
        if (++ltap == 9) { ltap = 0; ++lck; }                                                       \
        lslot = (lslot == 2) ? 0 : lslot + 1;                                                       \
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    C3D_STAMP(1);
    if (S > 0) {
        D3_LOAD_HALO(ck0);
        D3_DMA_W();              // the first two weight tiles fly while the group statistics are folded and the
        if (S > 1) D3_DMA_W();   // first halo chunk is normalised (the fold's plain LDS stores make the compiler drain
        if (FUSE) gn_fold_groups(p, b, gstat);   // them: everything requested so far lands together)
        __syncthreads();
        C3D_STAMP(2);
        D3_STORE_HALO(ck0);
    }
    C3D_STAMP(3);

    // fragment addressing, all of it hoisted out of the tap loop (the nine taps are unrolled: tap, ring slot and the halo
    // row / column shift are compile-time, a fragment address is a precomputed register -- or that register ^ 64 for the
    // second k-step -- plus an immediate): A chunk (kk, i) of tap (r, sx) = halo pixel (wm*4 + r + i) * HWD + frow + sx,
    // 6 rows x 3 shifts = 18 addresses per lane; the per-tap index arithmetic was ~60 VALU instructions of every step.
    unsigned ah[18];
#pragma unroll
    for (int ri = 0; ri < 6; ++ri)
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
            const int h_ = (wm * 4 + ri) * HWD + frow + sx;
            ah[ri * 3 + sx] = lds_base + (unsigned)(h_ * 8 + (fg ^ (h_ & 7))) * 16u;   // k-step 1: chunk fg + 4 = this ^ 64 bytes
        }
    const unsigned wf0 = wring + (unsigned)(((wn * (BN / 2) + frow) * 8 + (fg ^ (frow & 7))) * 16);   // + slot, + j * 2048; ^ 64
    const unsigned wf1 = wf0 ^ 64u;
    for (int ck = ck0; ck < ck1; ++ck) {
        const bool next_chunk = (ck + 1 < ck1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            constexpr int kW_BYTES = W_U4 * 16;
            const int cur = tap % 3;   // 9 taps per chunk: the ring slot of a tap is the same in every chunk
            // weight tile s has landed once at most the next tile's LW loads of this wave are outstanding (the halo loads
            // of tap 0 are issued BEFORE that tile's DMA, so they are older and covered by the same wait)
            C3D_STAMP(16 + ((ck - ck0) * 9 + tap) * 3);
            if (tap < 8 || next_chunk) c3_wait_vmcnt<LW>();
            else c3_wait_vmcnt<0>();
            C3D_STAMP(16 + ((ck - ck0) * 9 + tap) * 3 + 1);
            __builtin_amdgcn_s_barrier();   // this tap's tile (and at tap 0 the new halo) visible; ring slot (tap + 2) % 3 is free
            C3D_STAMP(16 + ((ck - ck0) * 9 + tap) * 3 + 2);
            if (tap == 0 && next_chunk) D3_LOAD_HALO(ck + 1);
            if (tap < 7 || next_chunk) D3_DMA_W();
            __builtin_amdgcn_sched_barrier(0);
            {
                const int r_ = tap / 3, sx_ = tap % 3;
                u32x4 af[2][MI], wf[2][NI];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(af[0][i]) : "v"(ah[(r_ + i) * 3 + sx_]));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[0][j]) : "v"(wf0), "n"(cur * kW_BYTES + j * 2048));
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(af[1][i]) : "v"(ah[(r_ + i) * 3 + sx_] ^ 64u));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[1][j]) : "v"(wf1), "n"(cur * kW_BYTES + j * 2048));
                c3_wait_lgkmcnt<MI + NI>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        mma16<T>(__builtin_bit_cast(uint4, wf[0][j]), __builtin_bit_cast(uint4, af[0][i]), acc[i][j]);
                c3_wait_lgkmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        mma16<T>(__builtin_bit_cast(uint4, wf[1][j]), __builtin_bit_cast(uint4, af[1][i]), acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (tap == 8 && next_chunk) {
                __builtin_amdgcn_s_barrier();   // single halo buffer: every wave has read its last fragments of this chunk
                D3_STORE_HALO(ck + 1);          // published by the barrier that opens the next tap
            }
        }
    }
#undef D3_LOAD_HALO
#undef D3_STORE_HALO
#undef D3_DMA_W
    C3D_STAMP(4);
    c3_wait_vmcnt<0>();
    __syncthreads();   // the LDS becomes the statistics scratch of the epilogue
    halo_tile_epilogue<T, BN>(p, acc, b, py0, px0, n0, z, reinterpret_cast<float*>(smem_raw));
    C3D_STAMP(5);
#ifdef C3D_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    C3D_STAMP(6);
#endif
#endif
}

template <typename T, int BN, bool FUSE>
int launch_dma_one(const IgemmP& p0, hipStream_t s) {
    IgemmP p = p0;
    constexpr size_t lds = (size_t)(HPIX * 8 + 3 * BN * 8) * 16 + 32 * sizeof(float2);
    auto kern = conv3x3_halo_dma_kernel<T, BN, FUSE>;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = madm_raise_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds), attr_done, "conv3x3 (LDS-DMA)")) return e;
    const int patchesX = (p.OW + TW - 1) / TW, patchesY = (p.OH + TH - 1) / TH;
    p.tilesN = (p.N + BN - 1) / BN;
    dim3 grid((unsigned)(p.B * patchesX * patchesY * p.tilesN), 1, (unsigned)p.splitk);
    PatchDecode pd;
    if (!patch_decode_fill(pd, patchesX, patchesY, p.tilesN, (long long)grid.x)) {
        madm_set_error("conv3x3 (LDS-DMA): grid of %u blocks too large for the reciprocal patch decode", grid.x);
        return MADM_ERR_INVALID_ARG;
    }
    kern<<<grid, 256, lds, s>>>(p, pd);
    return madm_check_launch("conv3x3_halo_dma_kernel");
}

}  // namespace

template <typename T>
int launch_conv3x3_halo_dma(const IgemmP& p, int bn, hipStream_t s) {
    const bool fuse = p.gn_sums1 != nullptr;
    if (bn == 128) return fuse ? launch_dma_one<T, 128, true>(p, s) : launch_dma_one<T, 128, false>(p, s);
    return fuse ? launch_dma_one<T, 64, true>(p, s) : launch_dma_one<T, 64, false>(p, s);
}
#ifdef C3D_STAMPS
extern "C" int madm_debug_read_c3d_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c3d_stamps), sizeof(unsigned long long) * n);
}
#endif
template int launch_conv3x3_halo_dma<float>(const IgemmP&, int, hipStream_t);
template int launch_conv3x3_halo_dma<bf16_t>(const IgemmP&, int, hipStream_t);
template int launch_conv3x3_halo_dma<f16_t>(const IgemmP&, int, hipStream_t);
