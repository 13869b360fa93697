// Shared by conv3x3.hip (register-staged weights) and conv3x3_dma.hip (LDS-DMA weights): patch geometry, the fused
// GroupNorm transform of one staged chunk and the patch epilogue.
#pragma once
#include "igemm_common.hpp"

// Block index -> (output-channel tile, image, patch row, patch column) without run-time integer divisions: the three divisors
// are launch constants, so the host hands over their reciprocals m = ceil(2^32 / d) and the kernel takes q = umulhi(x, m),
// exact while x * d < 2^32 (checked by the launcher); d == 1 passes m = 0 and means q = x.  (Three 32-bit divisions are ~120
// VALU instructions in front of the first DMA of every block: 2.2 k clocks of set-up in the 8 x 16 kernel's stamps, round 4.)
struct PatchDecode {
    int patchesX, patchesPerImg;
    unsigned m_tilesN, m_ppi, m_px;
};
__host__ inline unsigned patch_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }
__host__ inline bool patch_decode_fill(PatchDecode& pd, int patchesX, int patchesY, int tilesN, long long blocks) {
    pd.patchesX = patchesX; pd.patchesPerImg = patchesX * patchesY;
    pd.m_tilesN = patch_magic((unsigned)tilesN); pd.m_ppi = patch_magic((unsigned)pd.patchesPerImg);
    pd.m_px = patch_magic((unsigned)patchesX);
    const unsigned long long lim = 0x100000000ull;
    return (unsigned long long)blocks * (unsigned)tilesN < lim && (unsigned long long)blocks * (unsigned)pd.patchesPerImg < lim &&
           (unsigned long long)pd.patchesPerImg * (unsigned)patchesX < lim;
}
__device__ __forceinline__ int magic_div(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }

namespace {

constexpr int TH = 8, TW = 16, HWD = TW + 2, HPIX = (TH + 2) * HWD;  // 180 halo pixels
#ifdef C3_STAMPS
__device__ unsigned long long g_c3_stamps[2048];
#endif

template <typename T, int EPC>
__device__ __forceinline__ u32x4 gn_act_chunk(u32x4 raw, const float* sc, const float* sh, int act) {
    float f[EPC];
    chunk_to_f32<T>(__builtin_bit_cast(uint4, raw), f);
#pragma unroll
    for (int j = 0; j < EPC; ++j) f[j] = f[j] * sc[j] + sh[j];
    act_inplace<EPC>(f, act);
    return __builtin_bit_cast(u32x4, f32_to_chunk<T>(f));
}

// ---- patch epilogue shared by the register-staged and the LDS-DMA halo kernels ----
// lane: pixel = patch row wm*4+i, column frow; channels n .. n+3.  ``red`` = >= 4 * BN floats of LDS nobody reads any more.
template <typename T, int BN>
__device__ __forceinline__ void halo_tile_epilogue(const IgemmP& p, f32x4 (&acc)[4][BN / 32], int b, int py0, int px0,
                                                   int n0, int z, float* red) {
    constexpr int MI = 4, NI = BN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const bool want_stats = p.stats != nullptr && p.splitk == 1;
    const int nb = n0 + wn * (BN / 2) + fg * 4;
    f32x4 cs[NI], cq[NI], add[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { cs[j] = f32x4{0.f, 0.f, 0.f, 0.f}; cq[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (p.splitk == 1) epilogue_consts<NI>(p, nb, b, add);   // a patch lies in one image
    int mrow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int oy = py0 + wm * 4 + i, ox = px0 + frow;
        mrow[i] = (oy < p.OH && ox < p.OW) ? (b * p.OH + oy) * p.OW + ox : -1;
    }
    if (p.splitk > 1) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (mrow[i] < 0) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const f32x4 v = acc[i][j];
                if (nb + 16 * j < p.N)
                    *reinterpret_cast<float4*>(p.ws + ((size_t)z * p.M + mrow[i]) * p.N + nb + 16 * j) =
                        make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    } else {
        epilogue_tile<T, MI, NI>(p, mrow, nb, add, false, acc);
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
    }
    if (want_stats) {
        __syncthreads();
        for (int c = tid; c < 2 * BN; c += 256) {
            const int n = n0 + (c >> 1);
            if (n < p.N)
                atomicAdd(p.stats + ((size_t)b * p.N + n0) * 2 + c, (double)red[c] + (double)red[2 * BN + c]);
        }
    }
}

}  // namespace
