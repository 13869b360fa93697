// 3x3 / stride-1 / pad-1 convolution with the INPUT staged as an LDS halo tile, optionally fused with the
// GroupNorm(+SiLU) that precedes it in ResnetBlock2D (diffusers; reached from
// /root/reference/modeling/meta_arch/ldm_diffusers.py:290,387,435,609-611).
//
// One workgroup (256 threads, 4 waves as 2 x 2) computes an 8 x 16 pixel patch x BN output channels.
// Per 128-byte channel chunk the (8+2) x (16+2) halo of the patch is brought into LDS ONCE -- with
// y = silu(x * scale[b][c] + shift[b][c]) applied on the way when the GroupNorm is fused -- and all nine taps
// read it as shifted views; only the weight tile changes per tap.  Compared with the gather igemm this reads
// each input element ~1.4x instead of 9x per N-tile and removes the stand-alone GroupNorm-apply pass.
//   LDS: halo[2][180 px][8 x 16 B] + w[2][BN][8 x 16 B], chunk index XOR-swizzled with (row & 7);
//   pipeline: weights run two taps ahead in registers (as igemm.hip), the next chunk's halo is loaded at
//   tap 0 and written to the other halo buffer after tap 8; one barrier per tap.
// Epilogue (bias, time row, residual, GroupNorm sums of the output, split-K slabs) is shared with igemm.hip.
#include "conv3x3_common.hpp"

namespace {

template <typename T, int BN, bool FUSE, int NWS>
__global__ __launch_bounds__(256, (NWS == 2 || BN == 64) ? 2 : 1) void conv3x3_halo_kernel(const IgemmP p, const PatchDecode pd) {
    kernarg_touch<5>();
    constexpr int EPC = TT<T>::EPC;
    constexpr int BKE = 8 * EPC;
    constexpr int MI = 4, NI = BN / 32;     // wave tile: 4 patch rows (64 px) x BN/2 channels
    constexpr int RB = BN / 32;             // weight rows staged per thread
    constexpr int HI = (HPIX * 8 + 255) / 256;  // halo chunks staged per thread (6)
    constexpr int HALO_U4 = HPIX * 8, W_U4 = BN * 8;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u32x4* halo = reinterpret_cast<u32x4*>(smem_raw);   // [2][HALO_U4]
    u32x4* wlds = halo + 2 * HALO_U4;                   // [2][W_U4]
    float2* gstat = reinterpret_cast<float2*>(wlds + 2 * W_U4);   // [32] {mean, rstd} per group (FUSE)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int cpos = tid & 7, lrow = tid >> 3;
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

    // ---- per-thread staging state ----
    int pixoff[HI], hpos[HI];
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int idx = tid + 256 * i;
        const int h = idx >> 3;
        const int hy = h / HWD, hx = h - hy * HWD;
        const int iy = py0 - 1 + hy, ix = px0 - 1 + hx;
        const bool ok = idx < HPIX * 8 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        pixoff[i] = ok ? (b * p.IH + iy) * p.IW + ix : -1;
        hpos[i] = idx < HPIX * 8 ? h * 8 + (cpos ^ (h & 7)) : -1;
    }
    unsigned wvoff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + lrow + 32 * i;
        wvoff[i] = (n < p.N) ? (unsigned)(((size_t)n * p.ldw + cpos * EPC) * sizeof(T)) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in2 ? p.in2 : p.in1), 0,
                                                                         p.in2 ? p.bytes2 : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.bytesw, 0x00020000);

    const int nchunks = p.Ctot / BKE;
    const int ck0 = (nchunks * z) / p.splitk;
    const int ck1 = (nchunks * (z + 1)) / p.splitk;
    const int S = (ck1 - ck0) * 9;

    static_assert(NWS >= 2 && NWS % 2 == 0, "the weight ring must have an even number of stages");
    // rw[t % NWS] = weight tile t, NWS - 1 tiles ahead of the MFMAs.  Deeper rings (NWS = 4, 6) and a GroupNorm
    // transform spread over the taps were measured on MI355X and did not pay (round-1 notes in DESIGN.md): the step is
    // bound by its non-MFMA phases (LDS store, barrier), not by the weight latency.
    u32x4 hr[HI], rw[NWS][RB];
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) { sc[j] = 1.f; sh[j] = 0.f; }

#define C3_LOAD_HALO(ck)                                                                            \
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
            const float* gs_ = p.gn_gamma + c0_ + cpos * EPC;   /* raw gamma / beta; C3_STORE_HALO */     \
            const float* gh_ = p.gn_beta + c0_ + cpos * EPC;    /* turns them into scale / shift   */     \
            _Pragma("unroll") for (int j = 0; j < EPC; j += 4) {                                    \
                const float4 a_ = *reinterpret_cast<const float4*>(gs_ + j);                        \
                const float4 b_ = *reinterpret_cast<const float4*>(gh_ + j);                        \
                sc[j] = a_.x; sc[j + 1] = a_.y; sc[j + 2] = a_.z; sc[j + 3] = a_.w;                 \
                sh[j] = b_.x; sh[j + 1] = b_.y; sh[j + 2] = b_.z; sh[j + 3] = b_.w;                 \
            }                                                                                       \
        }                                                                                           \
    }

#define C3_STORE_HALO(buf, ck_)                                                                     \
    {                                                                                               \
        u32x4* dst_ = halo + (buf) * HALO_U4;                                                       \
        if (FUSE) {                                                                                 \
            const unsigned cb_ = (unsigned)((ck_) * BKE + cpos * EPC);                              \
            _Pragma("unroll") for (int j = 0; j < EPC; ++j) {                                       \
                const float2 st_ = gstat[__umulhi(cb_ + j, p.gn_magic)];                            \
                const float s_ = st_.y * sc[j];                                                     \
                sh[j] = sh[j] - st_.x * s_;                                                         \
                sc[j] = s_;                                                                         \
            }                                                                                       \
        }                                                                                           \
        _Pragma("unroll") for (int i = 0; i < HI; ++i) {                                            \
            if (hpos[i] >= 0) {                                                                     \
                u32x4 v_ = hr[i];                                                                   \
                if (FUSE) {                                                                         \
                    v_ = gn_act_chunk<T, EPC>(v_, sc, sh, p.act);                                   \
                    if (pixoff[i] < 0) v_ = u32x4{0u, 0u, 0u, 0u};   /* the conv pads the ACTIVATED tensor */ \
                }                                                                                   \
                dst_[hpos[i]] = v_;                                                                 \
            }                                                                                       \
        }                                                                                           \
    }

    // weight stream: (lck, ltap) = tile the next C3_LOAD_W fetches
    int lck = ck0, ltap = 0;
#define C3_LOAD_W(RW_)                                                                              \
    {                                                                                               \
        const unsigned kofs_ = (unsigned)(ltap * p.Ctot + lck * BKE) * (unsigned)sizeof(T);         \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                              \
            RW_[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff[i], kofs_, 0);                \
        if (++ltap == 9) { ltap = 0; ++lck; }                                                       \
    }
#define C3_STORE_W(buf, RW_)                                                                        \
    {                                                                                               \
        u32x4* dst_ = wlds + (buf) * W_U4;                                                          \
        const int sw_ = cpos ^ (lrow & 7);                                                          \
        _Pragma("unroll") for (int i = 0; i < RB; ++i) dst_[(lrow + 32 * i) * 8 + sw_] = RW_[i];    \
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (S > 0) {
        C3_LOAD_HALO(ck0);
        C3_LOAD_W(rw[0]);
        if constexpr (NWS == 2) {
            if (S > 1) C3_LOAD_W(rw[1]);
        } else {
#pragma unroll
            for (int u = 1; u < NWS; ++u)
                if (u < S) C3_LOAD_W(rw[u]);
        }
        if (FUSE) {   // GroupNorm finalize of this image while the first tiles are in flight
            gn_fold_groups(p, b, gstat);
            __syncthreads();
        }
        C3_STORE_HALO(0, ck0);
        C3_STORE_W(0, rw[0]);
    }
    __syncthreads();

#ifdef C3_EXP_NOPIN
#define C3_PIN()
#else
#define C3_PIN() __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef C3_EXP_SETPRIO
#define C3_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define C3_PRIO(x)
#endif
#ifdef C3_STAMPS   // tools/exp: shader-clock stamps of block 0 / wave 0 at four points of every step
#define C3_STAMP(s_, k_)                                                                            \
    if (blockIdx.x == 0 && blockIdx.z == 0 && tid == 0 && (s_) < 512)                               \
        g_c3_stamps[(s_) * 4 + (k_)] = __builtin_readcyclecounter();
#else
#define C3_STAMP(s_, k_)
#endif
    int ck = ck0, tap = 0;   // the tile being computed
#define C3_STEP(s_, CUR, RL_, RS_)                                                                  \
    {                                                                                               \
        const bool next_chunk_ = (ck + 1 < ck1);                                                    \
        C3_STAMP(s_, 0);                                                                            \
        if ((s_) + NWS < S) C3_LOAD_W(RL_);                                                           \
        if (tap == 0 && next_chunk_) C3_LOAD_HALO(ck + 1);                                          \
        C3_PIN();                                                                                   \
        C3_PRIO(1);                                                                                 \
        {                                                                                           \
            const u32x4* hsrc_ = halo + ((ck - ck0) & 1) * HALO_U4;                                 \
            const u32x4* wsrc_ = wlds + (CUR) * W_U4;                                               \
            const int r_ = tap / 3, sx_ = tap - 3 * r_;                                             \
            const int hb_ = (wm * 4 + r_) * HWD + frow + sx_;                                       \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                      \
                const int c_ = fg + 4 * kk;                                                         \
                uint4 af_[MI], wf_[NI];                                                             \
                _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                    \
                    const int h_ = hb_ + i * HWD;                                                   \
                    af_[i] = __builtin_bit_cast(uint4, hsrc_[h_ * 8 + (c_ ^ (h_ & 7))]);            \
                }                                                                                   \
                _Pragma("unroll") for (int j = 0; j < NI; ++j)                                      \
                    wf_[j] = __builtin_bit_cast(uint4, wsrc_[(wn * (BN / 2) + j * 16 + frow) * 8 + (c_ ^ (frow & 7))]); \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                      \
                    _Pragma("unroll") for (int j = 0; j < NI; ++j) mma16<T>(wf_[j], af_[i], acc[i][j]); \
            }                                                                                       \
        }                                                                                           \
        C3_PRIO(0);                                                                                 \
        C3_PIN();                                                                                   \
        C3_STAMP(s_, 1);                                                                            \
        if ((s_) + 1 < S) C3_STORE_W((CUR) ^ 1, RS_);                                               \
        if (tap == 8 && next_chunk_) C3_STORE_HALO(((ck - ck0) + 1) & 1, ck + 1);                   \
        C3_STAMP(s_, 2);                                                                            \
        __syncthreads();                                                                            \
        C3_STAMP(s_, 3);                                                                            \
        if (++tap == 9) { tap = 0; ++ck; }                                                          \
    }
    int s = 0;
    if constexpr (NWS == 2) {
        for (; s + 1 < S; s += 2) {
            C3_STEP(s, 0, rw[0], rw[1]);
            C3_STEP(s + 1, 1, rw[1], rw[0]);
        }
        if (s < S) C3_STEP(s, 0, rw[0], rw[1]);
    } else {
        for (; s + NWS <= S; s += NWS) {
#pragma unroll
            for (int u = 0; u < NWS; ++u) C3_STEP(s + u, u & 1, rw[u], rw[(u + 1) % NWS]);
        }
#pragma unroll
        for (int u = 0; u < NWS - 1; ++u)
            if (s + u < S) C3_STEP(s + u, u & 1, rw[u], rw[(u + 1) % NWS]);
    }
#undef C3_STEP
#undef C3_LOAD_W
#undef C3_STORE_W
#undef C3_LOAD_HALO
#undef C3_STORE_HALO

    halo_tile_epilogue<T, BN>(p, acc, b, py0, px0, n0, z, reinterpret_cast<float*>(smem_raw));
}

template <typename T, int BN, bool FUSE, int NWS>
int launch_one(const IgemmP& p0, hipStream_t s) {
    IgemmP p = p0;
    constexpr size_t lds = (size_t)(2 * HPIX * 8 + 2 * BN * 8) * 16 + 32 * sizeof(float2);
    auto kern = conv3x3_halo_kernel<T, BN, FUSE, NWS>;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = madm_raise_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds), attr_done, "conv3x3")) return e;
    const int patchesX = (p.OW + TW - 1) / TW, patchesY = (p.OH + TH - 1) / TH;
    p.tilesN = (p.N + BN - 1) / BN;
    dim3 grid((unsigned)(p.B * patchesX * patchesY * p.tilesN), 1, (unsigned)p.splitk);
    PatchDecode pd;
    if (!patch_decode_fill(pd, patchesX, patchesY, p.tilesN, (long long)grid.x)) {
        madm_set_error("conv3x3: grid of %u blocks too large for the reciprocal patch decode", grid.x);
        return MADM_ERR_INVALID_ARG;
    }
    kern<<<grid, 256, lds, s>>>(p, pd);
    return madm_check_launch("conv3x3_halo_kernel");
}

}  // namespace

template <typename T>
int launch_conv3x3_halo(const IgemmP& p, int bn, hipStream_t s) {
    const bool fuse = p.gn_sums1 != nullptr;
    if (bn == 128) return fuse ? launch_one<T, 128, true, 2>(p, s) : launch_one<T, 128, false, 2>(p, s);
    return fuse ? launch_one<T, 64, true, 2>(p, s) : launch_one<T, 64, false, 2>(p, s);
}
#ifdef C3_STAMPS
extern "C" int madm_debug_read_c3_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c3_stamps), sizeof(unsigned long long) * n);
}
#endif
template int launch_conv3x3_halo<float>(const IgemmP&, int, hipStream_t);
template int launch_conv3x3_halo<bf16_t>(const IgemmP&, int, hipStream_t);
template int launch_conv3x3_halo<f16_t>(const IgemmP&, int, hipStream_t);
