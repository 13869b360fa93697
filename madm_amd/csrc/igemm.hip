// Implicit-GEMM convolution / linear layer on gfx950 MFMA (madm_conv2d_fwd).
//
// GEMM view:  out[m][n] = sum_k A(m,k) * w[n][k],  m = (b, oy, ox), k = (kh, kw, c).
// Tile: BM x BN outputs per 256-thread workgroup (4 waves as 2 x 2), K tile = 128 bytes per row
// (64 bf16 / 32 f32).  Both operand tiles live in LDS as [row][8 x 16-byte chunks], chunk index
// XOR-swizzled with (row & 7) so the ds_read_b128 of "row = lane & 15, chunk = lane >> 4" is
// bank-conflict free.  Global -> register -> LDS staging, double buffered: the loads of K-tile
// t+1 are issued before the MFMAs of tile t and written to the other buffer after them; one
// barrier per K-tile.  The MFMA is issued as D = W_frag x A_frag so that every lane ends up with
// 4 CONSECUTIVE output channels of one pixel: 8/16-byte stores, vector bias / residual loads.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <atomic>
#include "igemm_common.hpp"

namespace {
#ifdef GLDS_STAMPS   // tools/exp/stamps_reg.py: block-level timeline of one block of the register-staged kernel
__device__ unsigned long long g_reg_stamps[64];
#define REG_BSTAMP(k_) if (blockIdx.x == 7 && blockIdx.z == 0 && threadIdx.x == 0) g_reg_stamps[(k_)] = __builtin_readcyclecounter();
#else
#define REG_BSTAMP(k_)
#endif

// Epilogue constants of a tile's BN columns (bias, and the time row when the tile lies in one image) are requested
// BEFORE the operand loads and parked in LDS: the epilogue starts with a ds_read instead of a global round trip
// (900 .. 1 200 of the 16 000 clocks of a 128 x 64 block at K = 320, in-kernel stamps).  Register-staged kernel: the first
// BN / 4 threads load float4s and write bias + row to the stash after the first tiles have been requested.
template <int BM, int BN>
__device__ __forceinline__ bool cstash_request(const IgemmP& p, int m0, int n0, u32x4& cb, u32x4& cr, u32x4& cl) {
    // buffer loads like the operand tiles' (ONE kind of memory operation in flight: the compiler can wait for these three
    // with a counted vmcnt instead of draining the tiles behind them); absent rows = zero-sized descriptors, lanes past
    // column N / past the first BN / 4 threads = out-of-range offsets: zeros, no branches
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x;
    const bool on = p.splitk == 1 && tid < BN / 4;
    const unsigned nbytes = (unsigned)p.N * 4u;
    const float* rowp = nullptr;
    if (p.rowvec) {
        const int OHW = p.OH * p.OW;
        int mlast = m0 + BM; if (mlast > p.M) mlast = p.M; mlast -= 1;
        const int img0 = m0 / OHW;
        if (mlast / OHW == img0) rowp = p.rowvec + (size_t)img0 * p.ldrv;
    }
    const float* dummy = reinterpret_cast<const float*>(p.w);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : dummy), 0, p.bias ? nbytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(rowp ? rowp : dummy), 0, rowp ? nbytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_cs ? p.ln_cs : dummy), 0, p.ln_cs ? nbytes : 0u, 0x00020000);
    const unsigned off = on ? (unsigned)(n0 + tid * 4) * 4u : OOB;
    cb = __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, 0);
    cr = __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 0);
    cl = __builtin_amdgcn_raw_buffer_load_b128(rl, off, 0, 0);   // folded LayerNorm: column sums of the weight
    return on;
}
template <int BN>
__device__ __forceinline__ void cstash_park(bool on, u32x4 cb_, u32x4 cr_, u32x4 cl_, float* cstash) {
    if (on) {   // published by the kernel's next barrier
        const float4 cb = __builtin_bit_cast(float4, cb_), cr = __builtin_bit_cast(float4, cr_);
        *reinterpret_cast<float4*>(cstash + threadIdx.x * 4) = make_float4(cb.x + cr.x, cb.y + cr.y, cb.z + cr.z, cb.w + cr.w);
        *reinterpret_cast<float4*>(cstash + BN + threadIdx.x * 4) = __builtin_bit_cast(float4, cl_);
    }
}

// ---- tile epilogue shared by the register-staged and the LDS-DMA kernels ----------------------------------------
// lane holds pixel m (lane & 15), channels n .. n+3 (4 * (lane >> 4)) of each 16x16 sub-tile.
// Optional fused GroupNorm statistics of the OUTPUT tensor: per-(image, channel) sum and sum of squares accumulated
// in the f64 stats[B][N][2] (f32 partials per block, f64 atomics across blocks: reproducible to f32 rounding whatever
// the arrival order) -- the consumer GroupNorm then needs no pass of its own.  ``red`` = >= 4 * BN floats of LDS that
// no wave reads any more (the caller has passed a barrier since the last MFMA operand read).
template <typename T, int BM, int BN>
__device__ __forceinline__ void igemm_tile_epilogue(const IgemmP& p, f32x4 (&acc)[BM / 32][BN / 32], int m0, int n0,
                                                    int z, float* red, float (&lns)[BM / 32], float (&lnq)[BM / 32],
                                                    const float* cstash, const float* cstash_row, const float* cstash_ln) {
    constexpr int MI = BM / 32, NI = BN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const int OHW = p.OH * p.OW;
    const bool want_stats = p.stats != nullptr && p.splitk == 1;
    int mlast = m0 + BM; if (mlast > p.M) mlast = p.M; mlast -= 1;
    const int img0 = m0 / OHW;
    const bool one_image = (mlast / OHW) == img0;   // block-uniform
    const int nb = n0 + wn * (BN / 2) + fg * 4;
    f32x4 cs[NI], cq[NI], add[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { cs[j] = f32x4{0.f, 0.f, 0.f, 0.f}; cq[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (p.splitk == 1) {   // bias (+ the image's time row when one_image), parked in LDS during the prologue; cstash_row:
                           // the LDS-DMA kernel parks the two rows separately (no adder on that path)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const float4 c = *reinterpret_cast<const float4*>(cstash + wn * (BN / 2) + fg * 4 + 16 * j);
            add[j] = f32x4{c.x, c.y, c.z, c.w};
            if (cstash_row) {
                const float4 r = *reinterpret_cast<const float4*>(cstash_row + wn * (BN / 2) + fg * 4 + 16 * j);
                add[j] += f32x4{r.x, r.y, r.z, r.w};
            }
        }
    }
    if (p.ln_cs) {
        // folded LayerNorm: the lane's partial row sums cover the chunks it read (lane >> 4, + 4 per half tile); the four
        // lanes that share lane & 15 hold the rest of the row
        f32x4 csv[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {   // parked in LDS with the other epilogue constants
            const float4 c = *reinterpret_cast<const float4*>(cstash_ln + wn * (BN / 2) + fg * 4 + 16 * j);
            csv[j] = f32x4{c.x, c.y, c.z, c.w};
        }
        const float inv_k = 1.0f / (float)p.K;
        // each of the two waves that own these rows (wn = 0 / 1) summed one K half of every tile: exchange through LDS
        // (behind the statistics scratch: a fast wave may already write that while a slow one still reads here)
        float2* lx = reinterpret_cast<float2*>(red + 4 * BN);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float sx = lns[i], sq = lnq[i];
            sx += __shfl_xor(sx, 16); sq += __shfl_xor(sq, 16);
            sx += __shfl_xor(sx, 32); sq += __shfl_xor(sq, 32);
            lns[i] = sx; lnq[i] = sq;
            if (fg == 0) lx[(wave * MI + i) * 16 + frow] = make_float2(sx, sq);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float2 o = lx[((wave ^ 1) * MI + i) * 16 + frow];
            const float sx = lns[i] + o.x, sq = lnq[i] + o.y;
            const float mean = sx * inv_k;
            float var = sq * inv_k - mean * mean;
            var = var < 0.f ? 0.f : var;
            const float rstd = 1.0f / sqrtf(var + p.ln_eps);
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = (acc[i][j] - mean * csv[j]) * rstd;
        }
    }
    REG_BSTAMP(8);
    int mrow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + frow;
        mrow[i] = m < p.M ? m : -1;
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
        epilogue_tile<T, MI, NI>(p, mrow, nb, add, !one_image, acc);
        if (want_stats) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (mrow[i] < 0) continue;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if (nb + 16 * j >= p.N) continue;
                    if (one_image) { cs[j] += acc[i][j]; cq[j] += acc[i][j] * acc[i][j]; }
                    else stats_add_elementwise(p, mrow[i], nb + 16 * j, acc[i][j]);
                }
            }
        }
    }
    REG_BSTAMP(9 + MI);
    if (want_stats && one_image) {
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
    if (want_stats && one_image) {
        __syncthreads();
        for (int c = tid; c < 2 * BN; c += 256) {   // c = channel * 2 + {sum, sumsq}
            const int n = n0 + (c >> 1);
            if (n < p.N) atomicAdd(p.stats + ((size_t)img0 * p.N + n0) * 2 + c, (double)red[c] + (double)red[2 * BN + c]);
        }
    }
}

template <typename T, int BM, int BN, int NST, bool LIN>   // LIN: the linear fast path, see igemm_glds_kernel
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? 1 : 2) void igemm_kernel(const IgemmP p) {
    kernarg_touch<5>();
    constexpr int EPC = TT<T>::EPC;
    constexpr int BKE = 8 * EPC;            // elements per K tile (128 B per row)
    constexpr int RA = BM / 32, RB = BN / 32;  // rows staged per thread
    constexpr int MI = BM / 32, NI = BN / 32;  // 16x16 sub-tiles per wave (wave tile BM/2 x BN/2)
    __shared__ __attribute__((aligned(16))) uint4 smem[2 * (BM + BN) * 8];
    __shared__ __attribute__((aligned(16))) float cstash[2 * BN];   // [bias + time row | LayerNorm column sums]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    REG_BSTAMP(0);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware order: blocks b and b+8 share an XCD (round-robin dispatch), so give every XCD a contiguous
    // range of tiles -- neighbouring output rows re-read the same input lines / weight panels from ITS L2.
    // (placement only changes speed; the remap is a bijection for any grid size)
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int tn = bid % p.tilesN, tm = bid / p.tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;
    const int chunk = tid & 7, lrow = tid >> 3;

    // ---- per-thread gather state: RA pixels of the A tile, RB rows of the weight tile ----
    // All global reads are raw buffer loads: an out-of-range voffset (padding pixels, rows beyond
    // M / N) returns zeros, so the gather is branch-free.
    constexpr unsigned OOB = 0x80000000u;
    int a_b[RA], a_iy[RA], a_ix[RA];
    const int OHW = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < p.M) {
            if constexpr (LIN) {
                a_b[i] = m; a_iy[i] = 0; a_ix[i] = 0;   // a_b = input row, a_iy >= 0 marks it valid
            } else {
                const int b = m / OHW;
                const int r = m - b * OHW;
                const int oy = r / p.OW;
                const int ox = r - oy * p.OW;
                a_b[i] = b * p.IH; a_iy[i] = oy * p.stride - p.pad_t; a_ix[i] = ox * p.stride - p.pad_l;
            }
        } else {
            a_b[i] = 0; a_iy[i] = -(1 << 24); a_ix[i] = 0;  // never in range
        }
    }
    unsigned wvoff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + lrow + 32 * i;
        wvoff[i] = (n < p.N) ? (unsigned)(((size_t)n * p.ldw + chunk * EPC) * sizeof(T)) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in2 ? p.in2 : p.in1), 0,
                                                                         p.in2 ? p.bytes2 : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.bytesw, 0x00020000);

    // (integer divisions by run-time values cost ~150 clocks each on this VALU and sit in front of the first load: the
    // common cases -- no split-K, linear layers -- take none)
    int kt0 = 0, kt1 = p.nk;
    if (p.splitk > 1) {
        kt0 = (p.nk * z) / p.splitk;
        kt1 = (p.nk * (z + 1)) / p.splitk;
    }
    // tap state of the NEXT tile to load
    int c0, tr, ts;
    if constexpr (LIN) {
        c0 = kt0 * BKE; tr = 0; ts = 0;      // one tap: k = channel
    } else {
        const int kbase = kt0 * BKE;
        const int tap = kbase / p.Ctot;
        c0 = kbase - tap * p.Ctot;
        tr = tap / p.KW;
        ts = tap - tr * p.KW;
    }
    const int IHe = p.upsample ? 2 * p.IH : p.IH;
    const int IWe = p.upsample ? 2 * p.IW : p.IW;
    const int ush = p.upsample ? 1 : 0;

    u32x4 ra[NST][RA], rb[NST][RB];   // NST register stages: the global loads run NST K-tiles ahead of the MFMAs

#define IGEMM_LOAD_TILE(kt, RA_, RB_)                                                                    \
    {                                                                                            \
        const bool first = c0 < p.C1;                                                            \
        const __amdgpu_buffer_rsrc_t rs = first ? rs1 : rs2;                                     \
        const int ld = first ? p.ld1 : p.ld2;                                                    \
        const int cofs = (first ? c0 : c0 - p.C1) + chunk * EPC;                                 \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                         \
            if constexpr (LIN) {                                                                 \
                const unsigned off = (unsigned)(a_b[i] * ld + cofs) * (unsigned)sizeof(T);       \
                RA_[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, a_iy[i] >= 0 ? off : OOB, 0, 0); \
            } else {                                                                             \
                const int iy = a_iy[i] + tr, ix = a_ix[i] + ts;                                  \
                const bool ok = (unsigned)iy < (unsigned)IHe && (unsigned)ix < (unsigned)IWe;    \
                const unsigned off = (unsigned)(((a_b[i] + (iy >> ush)) * p.IW + (ix >> ush)) * ld + cofs) * \
                                     (unsigned)sizeof(T);                                        \
                RA_[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0);        \
            }                                                                                    \
        }                                                                                        \
        const unsigned kofs = (unsigned)(kt) * (unsigned)(BKE * sizeof(T));                      \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                           \
            RB_[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff[i], kofs, 0);              \
        c0 += BKE;                                                                               \
        if (c0 >= p.Ctot) { c0 = 0; ++ts; if (ts == p.KW) { ts = 0; ++tr; } }                    \
    }

#define IGEMM_STORE_TILE(buf, RA_, RB_)                                                                  \
    {                                                                                            \
        u32x4* sA = reinterpret_cast<u32x4*>(smem) + (buf) * (BM + BN) * 8;                      \
        u32x4* sB = sA + BM * 8;                                                                 \
        const int sw = chunk ^ (lrow & 7);                                                       \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) sA[(lrow + 32 * i) * 8 + sw] = RA_[i];    \
        _Pragma("unroll") for (int i = 0; i < RB; ++i) sB[(lrow + 32 * i) * 8 + sw] = RB_[i];    \
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float lns[MI], lnq[MI];   // folded LayerNorm: partial row sums of the lane's A-fragment rows
#pragma unroll
    for (int i = 0; i < MI; ++i) { lns[i] = 0.f; lnq[i] = 0.f; }
    const bool ln_on = p.ln_cs != nullptr;   // block-uniform

    // Pipeline: LDS buffer ((t - kt0) & 1) holds tile t; register stage (t + j) % NST holds tile t + j for
    // j = 1 .. NST-1; the loads of tile t + NST are issued before the MFMAs of tile t into the stage that held
    // tile t, and get NST MFMA phases to land (small grids have no co-resident block to hide the latency).
    REG_BSTAMP(1);
    u32x4 cst_b, cst_r, cst_l;
    const bool cst_on = cstash_request<BM, BN>(p, m0, n0, cst_b, cst_r, cst_l);
#pragma unroll
    for (int u = 0; u < NST; ++u)
        if (kt0 + u < kt1) IGEMM_LOAD_TILE(kt0 + u, ra[u], rb[u]);
    cstash_park<BN>(cst_on, cst_b, cst_r, cst_l, cstash);
    if (kt0 < kt1) IGEMM_STORE_TILE(0, ra[0], rb[0]);
    __syncthreads();

    const int frow = lane & 15, fg = lane >> 4, fsw = lane & 7;
#define IGEMM_COMPUTE(cur)                                                                          \
    {                                                                                               \
        const uint4* sA = smem + (cur) * (BM + BN) * 8;                                             \
        const uint4* sB = sA + BM * 8;                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                          \
            const int c = (fg + 4 * kk) ^ fsw;                                                      \
            uint4 af[MI], wf[NI];                                                                   \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                          \
                af[i] = sA[(wm * (BM / 2) + i * 16 + frow) * 8 + c];                                \
            _Pragma("unroll") for (int j = 0; j < NI; ++j)                                          \
                wf[j] = sB[(wn * (BN / 2) + j * 16 + frow) * 8 + c];                                \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                          \
                _Pragma("unroll") for (int j = 0; j < NI; ++j) mma16<T>(wf[j], af[i], acc[i][j]);   \
            if (ln_on && kk == wn) {   /* the two waves that share these rows split the K halves */ \
                _Pragma("unroll") for (int i = 0; i < MI; ++i) ln_accum<T>(af[i], lns[i], lnq[i]);  \
            }                                                                                       \
        }                                                                                           \
    }
    REG_BSTAMP(2);
    for (int kt = kt0; kt < kt1; kt += NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int t = kt + u;
            if (t < kt1) {
                const int cur = (t - kt0) & 1;
                if (t + NST < kt1) IGEMM_LOAD_TILE(t + NST, ra[u], rb[u]);            // stage u held tile t (in LDS now)
                __builtin_amdgcn_sched_barrier(0);
                IGEMM_COMPUTE(cur);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < kt1) IGEMM_STORE_TILE(cur ^ 1, ra[(u + 1) % NST], rb[(u + 1) % NST]);
                __syncthreads();
            }
        }
    }
#undef IGEMM_COMPUTE
#undef IGEMM_LOAD_TILE
#undef IGEMM_STORE_TILE

    REG_BSTAMP(3);
    igemm_tile_epilogue<T, BM, BN>(p, acc, m0, n0, z, reinterpret_cast<float*>(smem), lns, lnq, cstash, nullptr, cstash + BN);
    REG_BSTAMP(4);
#ifdef GLDS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    REG_BSTAMP(5);
#endif
}

// ---- LDS-DMA variant ----------------------------------------------------------------------------------------------
// Same GEMM view, tiles and LDS image as igemm_kernel, but both operand tiles travel global -> LDS directly
// (`buffer_load_dwordx4 ... lds`: each wave instruction lands 8 rows x 128 B at M0 + lane * 16): no staging registers,
// no ds_write phase, NS - 1 K-tiles in flight per block.  The XOR swizzle is applied on the SOURCE side (lane (r, p)
// fetches global chunk p ^ (r & 7)), so the LDS image -- and the fragment reads -- are those of igemm_kernel.
// Per K-tile: counted s_waitcnt vmcnt (only this tile's loads must have landed) -> one barrier (tile visible to every
// wave, previous slot free) -> issue the loads of tile t + NS - 1 into that slot -> ds_read + MFMA.
#define GLDS_ASM(...) asm volatile(__VA_ARGS__)
#ifdef GLDS_STAMPS
__device__ unsigned long long g_glds_stamps[2048];
#endif
#if defined(__HIP_DEVICE_COMPILE__)
template <int N> __device__ __forceinline__ void wait_vmcnt() { GLDS_ASM("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { GLDS_ASM("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#endif

// LIN: 1x1 / stride 1 / no padding / no upsampling (every linear layer and 1x1 conv): output row m reads input row m, so
// the gather needs no pixel arithmetic -- neither in the set-up (two integer divisions per staged row) nor per DMA
// instruction (the address is row * ld + channel offset).
template <typename T, int BM, int BN, int NS, bool LIN>
__global__ __launch_bounds__(256, BM * BN >= 128 * 128 ? (NS == 2 ? 2 : 1) : (NS == 3 && BM == 64 ? 3 : 2)) void igemm_glds_kernel(const IgemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)   // LDS address-space casts and gfx asm: device pass only (the host needs the stub)
    kernarg_touch<5>();
    constexpr int EPC = TT<T>::EPC;
    constexpr int BKE = 8 * EPC;
    constexpr int MI = BM / 32, NI = BN / 32;
    constexpr int ROWS = BM + BN;
    constexpr int LPT = ROWS / 32;          // wave instructions (8 rows each) per wave and K-tile
    constexpr int TILE_U4 = ROWS * 8;
    // NS = 2: the load of tile t + 1 is issued behind the barrier of step t and must have landed at step t + 1 (vmcnt(0)):
    // it overlaps this block's MFMAs of tile t only -- for tiles whose two resident blocks cover each other (128 x 128)
    static_assert(NS >= 2 && NS <= 4 && ROWS % 32 == 0, "ring depth / tile shape");
    extern __shared__ __attribute__((aligned(16))) uint4 gsmem[];   // [NS][ROWS][8 x 16 B]
    __shared__ __attribute__((aligned(16))) float cstash[3 * 256];   // three wave-wide DMA rows: bias, time row, LayerNorm column sums
    typedef __attribute__((address_space(3))) char lds_char;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
#ifdef GLDS_STAMPS
#define GLDS_BSTAMP(k_) if (blockIdx.x == 0 && blockIdx.z == 0 && tid == 0) g_glds_stamps[2000 + (k_)] = __builtin_readcyclecounter();
#else
#define GLDS_BSTAMP(k_)
#endif
    GLDS_BSTAMP(0);
    int bid = blockIdx.x;   // XCD-aware order, see igemm_kernel
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int tn = bid % p.tilesN, tm = bid / p.tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;
    const int rsub = lane >> 3, cpos = lane & 7;

    constexpr unsigned OOB = 0x80000000u;
    const int OHW = p.OH * p.OW;
    // per wave instruction i: rows 8 * (wave * LPT + i) .. + 7 of the stacked [A | W] tile
    int a_b[LPT], a_iy[LPT], a_ix[LPT];
    unsigned wvoff[LPT], swz[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int row = (wave * LPT + i) * 8 + rsub;
        swz[i] = (unsigned)((cpos ^ (row & 7)) * EPC);
        a_b[i] = 0; a_iy[i] = -(1 << 24); a_ix[i] = 0; wvoff[i] = OOB;
        if (row < BM) {
            const int m = m0 + row;
            if (m < p.M) {
                if constexpr (LIN) {
                    a_b[i] = m; a_iy[i] = 0;   // a_b = input row, a_iy >= 0 marks it valid
                } else {
                    const int b = m / OHW;
                    const int r = m - b * OHW;
                    const int oy = r / p.OW;
                    const int ox = r - oy * p.OW;
                    a_b[i] = b * p.IH; a_iy[i] = oy * p.stride - p.pad_t; a_ix[i] = ox * p.stride - p.pad_l;
                }
            }
        } else {
            const int n = n0 + row - BM;
            if (n < p.N) wvoff[i] = (unsigned)(((size_t)n * p.ldw + swz[i]) * sizeof(T));
        }
    }
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in2 ? p.in2 : p.in1), 0,
                                                                         p.in2 ? p.bytes2 : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.bytesw, 0x00020000);

    int kt0 = 0, kt1 = p.nk;     // (no run-time divisions in front of the first load in the common cases, see igemm_kernel)
    if (p.splitk > 1) {
        kt0 = (p.nk * z) / p.splitk;
        kt1 = (p.nk * (z + 1)) / p.splitk;
    }
    const int nt = kt1 - kt0;
    int c0, tr, ts;   // tap state of the NEXT tile to load
    if constexpr (LIN) {
        c0 = kt0 * BKE; tr = 0; ts = 0;
    } else {
        const int kbase = kt0 * BKE;
        const int tap = kbase / p.Ctot;
        c0 = kbase - tap * p.Ctot;
        tr = tap / p.KW;
        ts = tap - tr * p.KW;
    }
    const int IHe = p.upsample ? 2 * p.IH : p.IH;
    const int IWe = p.upsample ? 2 * p.IW : p.IW;
    const int ush = p.upsample ? 1 : 0;
    lds_char* const lds0 = (lds_char*)gsmem;

#define GLDS_LOAD_TILE(kt, slot)                                                                                 \
    {                                                                                                            \
        const bool first = c0 < p.C1;                                                                            \
        const __amdgpu_buffer_rsrc_t rs = first ? rs1 : rs2;                                                     \
        const int ld = first ? p.ld1 : p.ld2;                                                                    \
        const int cbase = first ? c0 : c0 - p.C1;                                                                \
        const unsigned kofs = (unsigned)(kt) * (unsigned)(BKE * sizeof(T));                                      \
        _Pragma("unroll") for (int i = 0; i < LPT; ++i) {                                                        \
            const int q = wave * LPT + i;                                                                        \
            lds_char* dst = lds0 + ((slot) * TILE_U4 + q * 64) * 16;                                             \
            if (q * 8 < BM) {   /* wave-uniform */                                                               \
                if constexpr (LIN) {                                                                             \
                    const unsigned off = (unsigned)(a_b[i] * ld + cbase + (int)swz[i]) * (unsigned)sizeof(T);    \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, a_iy[i] >= 0 ? off : OOB, 0, 0, 0);    \
                } else {                                                                                         \
                    const int iy = a_iy[i] + tr, ix = a_ix[i] + ts;                                              \
                    const bool ok = (unsigned)iy < (unsigned)IHe && (unsigned)ix < (unsigned)IWe;                \
                    const unsigned off = (unsigned)(((a_b[i] + (iy >> ush)) * p.IW + (ix >> ush)) * ld + cbase + \
                                                    (int)swz[i]) * (unsigned)sizeof(T);                          \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, ok ? off : OOB, 0, 0, 0);              \
                }                                                                                                \
            } else {                                                                                             \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dst, 16, wvoff[i], kofs, 0, 0);                    \
            }                                                                                                    \
        }                                                                                                        \
        c0 += BKE;                                                                                               \
        if (c0 >= p.Ctot) { c0 = 0; ++ts; if (ts == p.KW) { ts = 0; ++tr; } }                                    \
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float lns[MI], lnq[MI];   // folded LayerNorm: partial row sums of the lane's A-fragment rows
#pragma unroll
    for (int i = 0; i < MI; ++i) { lns[i] = 0.f; lnq[i] = 0.f; }
    const bool ln_on = p.ln_cs != nullptr;   // block-uniform

    GLDS_BSTAMP(1);
    // Epilogue constants: wave 0 fetches the tile's bias row, (tile in one image) time row and LayerNorm column sums straight
    // into the stash by LDS-DMA -- three more pieces, OLDER than every piece of the ring (loads retire in order: the ring's counted waits hold),
    // no register and no compiler-visible LDS write (which would make the compiler drain the ring).  Lanes past column
    // N, and absent rows (zero-sized descriptor), deliver zeros.
    if (p.splitk == 1 && wave == 0) {
        const unsigned nbytes = p.N > n0 ? (unsigned)(p.N - n0) * 4u : 0u;
        const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.bias ? p.bias + n0 : (const float*)p.w), 0, p.bias ? nbytes : 0u, 0x00020000);
        const float* rowp = nullptr;
        if (p.rowvec) {
            int mlast = m0 + BM; if (mlast > p.M) mlast = p.M; mlast -= 1;
            const int img0 = m0 / OHW;
            if (mlast / OHW == img0) rowp = p.rowvec + (size_t)img0 * p.ldrv + n0;
        }
        const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(rowp ? rowp : (const float*)p.w), 0, rowp ? nbytes : 0u, 0x00020000);
        const unsigned off = lane < BN / 4 ? (unsigned)lane * 16u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_char*)cstash, 16, off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsr, (lds_char*)cstash + 1024, 16, off, 0, 0, 0);
        const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.ln_cs ? p.ln_cs + n0 : (const float*)p.w), 0, p.ln_cs ? nbytes : 0u, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsl, (lds_char*)cstash + 2048, 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (u < nt) GLDS_LOAD_TILE(kt0 + u, u);
    GLDS_BSTAMP(2);

    const int frow = lane & 15, fg = lane >> 4, fsw = lane & 7;
    unsigned aA[2], aB[2];   // LDS byte addresses of this lane's fragment chunk in slot 0 (rows + i * 16 via offset:)
    {
        const unsigned base = (unsigned)(size_t)lds0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = (fg + 4 * kk) ^ fsw;
            aA[kk] = base + (unsigned)(((wm * (BM / 2) + frow) * 8 + c) * 16);
            aB[kk] = base + (unsigned)(((BM + wn * (BN / 2) + frow) * 8 + c) * 16);
        }
    }
    // Measured and rejected (MI355X, in-kernel stamps tools/exp/stamps_glds.py): the 4 DMA instructions of a 64x64 tile
    // cost ~550 of the ~1250 clocks of a K-step (the CU's load path moves 64 B/clk and both resident blocks issue right
    // after their barrier); issuing them between groups of MFMAs did not hide that (+9 % time), and a 128x128 tile
    // (one block per CU) was 1.5x slower than the register-staged 128x64 kernel.
    // one K-tile; CUR / NXT are compile-time slots (the loop is unrolled by NS) so the compiler can tell the slot being
    // read from the slots the in-flight DMA writes, and does not drain vmcnt before the ds_reads
#ifdef GLDS_STAMPS   // tools/exp: shader-clock stamps of block 0 / wave 0 (step start, DMA landed, barrier passed, MFMAs issued)
#define GLDS_STAMP(t_, k_)                                                                                       \
    if (blockIdx.x == 0 && blockIdx.z == 0 && tid == 0 && (t_) < 500) g_glds_stamps[(t_) * 4 + (k_)] = __builtin_readcyclecounter();
#else
#define GLDS_STAMP(t_, k_)
#endif
#define GLDS_STEP(t, CUR, NXT)                                                                                   \
    {                                                                                                            \
        const int ahead = (nt - 1 - (t) < NS - 2) ? nt - 1 - (t) : NS - 2;   /* tiles issued after tile t */       \
        GLDS_STAMP(t, 0);                                                                                        \
        if (ahead >= 2) wait_vmcnt<2 * LPT>();                                                                   \
        else if (ahead == 1) wait_vmcnt<LPT>();                                                                  \
        else wait_vmcnt<0>();                                                                                    \
        GLDS_STAMP(t, 1);                                                                                        \
        __builtin_amdgcn_s_barrier();   /* tile t landed for every wave; slot NXT (tile t - 1) is free */        \
        GLDS_STAMP(t, 2);                                                                                        \
        if ((t) + NS - 1 < nt) GLDS_LOAD_TILE(kt0 + (t) + NS - 1, NXT);                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        {                                                                                                        \
            /* fragment reads as inline asm: the compiler drains vmcnt(0) before any ds_read it can see while   \
               LDS-DMA is in flight (it cannot tell the slots apart), which would serialise the ring */         \
            u32x4 af[2][MI], wf[2][NI];                                                                          \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                   \
                const unsigned pa = aA[kk] + (CUR) * (TILE_U4 * 16), pb = aB[kk] + (CUR) * (TILE_U4 * 16);       \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                   \
                    GLDS_ASM("ds_read_b128 %0, %1 offset:%2" : "=v"(af[kk][i]) : "v"(pa), "n"(i * 2048));         \
                _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                   \
                    GLDS_ASM("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[kk][j]) : "v"(pb), "n"(j * 2048));         \
            }                                                                                                    \
            wait_lgkmcnt<MI + NI>();                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                       \
                _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                   \
                    mma16<T>(__builtin_bit_cast(uint4, wf[0][j]), __builtin_bit_cast(uint4, af[0][i]), acc[i][j]); \
            wait_lgkmcnt<0>();                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                       \
                _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                   \
                    mma16<T>(__builtin_bit_cast(uint4, wf[1][j]), __builtin_bit_cast(uint4, af[1][i]), acc[i][j]); \
            if (ln_on) {   /* the two waves that share these rows (wn = 0 / 1) split the K halves */              \
                if (wn == 0) {                                                                                   \
                    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                               \
                        ln_accum<T>(__builtin_bit_cast(uint4, af[0][i]), lns[i], lnq[i]);                        \
                } else {                                                                                         \
                    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                               \
                        ln_accum<T>(__builtin_bit_cast(uint4, af[1][i]), lns[i], lnq[i]);                        \
                }                                                                                                \
            }                                                                                                    \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        GLDS_STAMP(t, 3);                                                                                        \
    }
    int t = 0;
    for (; t + NS <= nt; t += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) GLDS_STEP(t + u, u, (u + NS - 1) % NS);
    }
#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (t + u < nt) GLDS_STEP(t + u, u, (u + NS - 1) % NS);
#undef GLDS_STEP
#undef GLDS_LOAD_TILE
    GLDS_BSTAMP(3);
    __syncthreads();   // every wave is done with the last tile: the LDS becomes the statistics scratch
    igemm_tile_epilogue<T, BM, BN>(p, acc, m0, n0, z, reinterpret_cast<float*>(gsmem), lns, lnq, cstash, cstash + 256, cstash + 512);
    GLDS_BSTAMP(4);
#endif
}

// sums the split-K slabs and applies the epilogue (+ the fused GroupNorm statistics).
// Block = 16 channel quads x 16 rows (one row per thread): a 16-row x 64-channel output tile.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmP p) {
    __shared__ float red[16][64][2];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int n = (blockIdx.x * 16 + tx) * 4;
    const int mbase = blockIdx.y * 16;
    const int OHW = p.OH * p.OW;
    int mlast = mbase + 16; if (mlast > p.M) mlast = p.M; mlast -= 1;
    const int img0 = mbase / OHW;
    const bool one_image = (mlast / OHW) == img0;
    const bool want_stats = p.stats != nullptr;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m = mbase + ty;
    const bool valid = m < p.M && n < p.N;
    if (valid) {
        // everything the epilogue adds is requested BEFORE the slab round trips (it used to follow them: one more
        // dependent L2 latency in a kernel that is nothing but latencies)
        const bool plain = p.epilogue != MADM_EPI_GEGLU;
        f32x4 add = f32x4{0.f, 0.f, 0.f, 0.f}, res = f32x4{0.f, 0.f, 0.f, 0.f};
        if (plain) {
            if (p.bias) {
                const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                add = f32x4{b.x, b.y, b.z, b.w};
            }
            if (p.rowvec) {
                const float4 r = *reinterpret_cast<const float4*>(p.rowvec + (size_t)(m / OHW) * p.ldrv + n);
                add[0] += r.x; add[1] += r.y; add[2] += r.z; add[3] += r.w;
            }
            if (p.residual) res = load4<T>(reinterpret_cast<const T*>(p.residual) + (size_t)m * p.ldr + n);
        }
        const float* src = p.ws + (size_t)m * p.N + n;
        const size_t slab = (size_t)p.M * p.N;
        // slabs are summed in slab order (bit-reproducible), eight loads in flight per round
        for (int zz = 0; zz < p.splitk; zz += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t[u] = (zz + u < p.splitk) ? *reinterpret_cast<const float4*>(src + (size_t)(zz + u) * slab)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (zz + u < p.splitk) { v[0] += t[u].x; v[1] += t[u].y; v[2] += t[u].z; v[3] += t[u].w; }
        }
        if (plain) {
            v += add;
            if (p.residual) v += res;
            if (p.epilogue == MADM_EPI_RELU) {
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            }
            if (p.out_f32) store4<float>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n, v);
            else store4<T>(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldo + n, v);
        } else {
            v = epilogue_store<T>(p, m, n, v);
        }
        if (want_stats && !one_image) stats_add_elementwise(p, m, n, v);
    } else {
        v = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (want_stats && one_image) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { red[ty][tx * 4 + r][0] = v[r]; red[ty][tx * 4 + r][1] = v[r] * v[r]; }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int c = threadIdx.x >> 1, w = threadIdx.x & 1;
            const int nn = blockIdx.x * 64 + c;
            if (nn < p.N) {
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) a += (double)red[k][c][w];
                atomicAdd(p.stats + ((size_t)img0 * p.N + nn) * 2 + w, a);
            }
        }
    }
}

// Split-K reduction + the GroupNorm(+act) that consumes the conv (madm_conv2d_args.pn_gamma; ResnetBlock2D conv1 -> norm2 ->
// SiLU).  One workgroup per (group, image): the slabs of the group's HW x cpg values are summed (in slab order, then + bias
// + time row: the order of splitk_reduce_kernel) into LDS as f32 pairs; mean / variance come from those f32 values (f32
// partials of <= 2 * ceil(units / 1024) values per thread, f64 across threads; the algebra of gn_apply_kernel); the second
// pass normalises out of LDS.  The raw conv output never exists in memory.  Everything it reads was written by the launch
// before it (L2-resident slabs): a latency kernel, so 16 waves per workgroup with four slab loads in flight each.
struct PostGn { const float* gamma; const float* beta; int G; float eps; int act; };
constexpr int PGN_THREADS = 1024;
constexpr size_t PGN_MAX_LDS = 96 * 1024;

// V = floats per unit (4 when the group's channel count allows 16-byte accesses, else 2).
template <typename T, int V>
__global__ __launch_bounds__(PGN_THREADS) void splitk_groupnorm_kernel(const IgemmP p, const PostGn pn) {
    extern __shared__ __attribute__((aligned(16))) float pgn_vals[];
    __shared__ double pgn_red[2][PGN_THREADS / 64];
    __shared__ float pgn_mr[2];
    typedef float vec __attribute__((ext_vector_type(V)));
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = p.N / pn.G, upp = cpg / V;           // units per pixel
    const int HW = p.OH * p.OW;
    const int units = HW * upp;
    const size_t slab = (size_t)p.M * p.N;
    const float* base = p.ws + (size_t)b * HW * p.N + g * cpg;
    float s = 0.f, q = 0.f;
    // 64 workgroups feed on slabs all 256 CUs just wrote: what bounds this loop is the bytes one CU keeps in flight, so
    // every thread requests eight slabs of its unit at once (16 waves x 64 lanes x 8 x 16 B = 128 KB per CU)
    constexpr int SF = 8;
    for (int u = threadIdx.x; u < units; u += PGN_THREADS) {
        const int px = u / upp, c = (u - px * upp) * V;
        const float* src = base + (size_t)px * p.N + c;
        vec add = {};
        if (p.bias) add = *reinterpret_cast<const vec*>(p.bias + g * cpg + c);
        if (p.rowvec) add += *reinterpret_cast<const vec*>(p.rowvec + (size_t)b * p.ldrv + g * cpg + c);
        vec v = {};
        for (int z = 0; z < p.splitk; z += SF) {
            vec t[SF];
#pragma unroll
            for (int k = 0; k < SF; ++k) {
                t[k] = vec{};
                if (z + k < p.splitk) t[k] = *reinterpret_cast<const vec*>(src + (size_t)(z + k) * slab);
            }
#pragma unroll
            for (int k = 0; k < SF; ++k)
                if (z + k < p.splitk) v += t[k];         // slab order: the sums of splitk_reduce_kernel, bit for bit
        }
        v += add;
        *reinterpret_cast<vec*>(pgn_vals + (size_t)u * V) = v;
#pragma unroll
        for (int j = 0; j < V; ++j) { s += v[j]; q = fmaf(v[j], v[j], q); }
    }
    double ds = (double)s, dq = (double)q;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { ds += __shfl_xor(ds, o); dq += __shfl_xor(dq, o); }
    if ((threadIdx.x & 63) == 0) { pgn_red[0][threadIdx.x >> 6] = ds; pgn_red[1][threadIdx.x >> 6] = dq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = 0.0, Q = 0.0;
#pragma unroll
        for (int w = 0; w < PGN_THREADS / 64; ++w) { S += pgn_red[0][w]; Q += pgn_red[1][w]; }
        const double inv_cnt = 1.0 / ((double)HW * (double)cpg);
        const double mean = S * inv_cnt;
        double var = Q * inv_cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        pgn_mr[0] = (float)mean;
        pgn_mr[1] = __builtin_amdgcn_rsqf((float)var + pn.eps);
    }
    __syncthreads();
    const float mean = pgn_mr[0], rstd = pgn_mr[1];
    T* ob = reinterpret_cast<T*>(p.out) + (size_t)b * HW * p.ldo + g * cpg;
    for (int u = threadIdx.x; u < units; u += PGN_THREADS) {
        const int px = u / upp, c = (u - px * upp) * V;
        const vec gm = *reinterpret_cast<const vec*>(pn.gamma + g * cpg + c);
        const vec bt = *reinterpret_cast<const vec*>(pn.beta + g * cpg + c);
        const vec v = *reinterpret_cast<const vec*>(pgn_vals + (size_t)u * V);
        float y[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {   // scale / shift formed as in gn_apply_kernel: x (rstd gamma) + (beta - mean rstd gamma)
            const float sc = rstd * gm[j];
            y[j] = act_f(v[j] * sc + (bt[j] - mean * sc), pn.act);
        }
        T* o = ob + (size_t)px * p.ldo + c;
        if constexpr (V == 4) store4<T>(o, f32x4{y[0], y[1], y[2], y[3]});
        else store2<T>(o, y[0], y[1]);
    }
}

// can the split-K reduction of this launch apply the consumer's GroupNorm?  (mirrors the clamps of madm_conv2d_fwd)
inline size_t post_gn_lds(int HW, int N, int G) { return (size_t)HW * (size_t)(N / G) * sizeof(float); }

int g_tile_override = 0;  // 0 = tuned table then heuristic; -1 = heuristic only; 1..6 = forced tile code
inline bool is_igemm_tile(int t) { return t <= 3 || (t >= 6 && t <= 8) || t == 11 || (t >= 14 && t <= 17); }
// tile 13 (igemm_apanel.hip): plain linear layer, one source, whole rows resident: no split-K, no residual / time row /
// fused output statistics (its epilogue touches no global memory but the stores)
inline bool apanel_eligible(const madm_conv2d_args* a) {
    return a->KH == 1 && a->KW == 1 && a->stride == 1 && a->pad_t == 0 && a->pad_l == 0 && !a->upsample && a->C2 == 0 &&
           a->OH == a->IH && a->OW == a->IW && a->splitk <= 1 && !a->stats && !a->residual && !a->rowvec && !a->gn_sums1 &&
           igemm_apanel_bm(a->C1, (int)madm_esize(a->dtype)) > 0 &&
           // its stores go through a buffer descriptor with 32-bit offsets (0x80000000 = "drop this lane")
           (size_t)a->B * a->OH * a->OW * (size_t)a->ldo * (a->out_f32 ? 4 : madm_esize(a->dtype)) < 0x80000000ull;
}
inline bool is_halo_tile(int t) { return t == 4 || t == 5 || t == 9 || t == 10 || t == 12; }

// Launch configurations measured on MI355X by tools/tune_insitu.py for the layer shapes of the SD-v1-4
// feature extractor at bs=2, 512x512 (any other shape falls back to the heuristics below).
// variant: 0 = plain, 1 = GroupNorm fused into the halo load, 2 = nearest-2x upsample gather, 3 = stride 2 (a downsample
// conv shares M, N, K with a stride-1 conv of the next level: 8 x 8 x 1280 of the UNet; without a row of its own it takes
// the plain row)
struct Tuned { int dtype, M, N, K, KH, variant, tile, splitk; };
inline int variant_of(const madm_conv2d_args* a) { return a->gn_sums1 ? 1 : (a->upsample ? 2 : (a->stride == 2 ? 3 : 0)); }
const Tuned g_tuned[] = {
#include "igemm_tuned.inc"
    {-1, 0, 0, 0, 0, 0, 0, 0}};
// The table above is tuned for THROUGHPUT: rows chosen with three launches of the layer side by side (tools/tune_concurrent.py), the
// neighbours a launch has under the runners of madm_amd/pipeline.py.  A synchronous caller -- the reference's loop calling forward()
// with one batch in flight -- wants the choice that is fastest ALONE on an idle chip: more split-K, the tile that fills 256 CUs by
// itself.  Profile 1 (madm_set_tuning_profile; ops.tuning_profile("latency")) puts these rows in front of the table; a shape without
// one keeps its throughput row.
const Tuned g_tuned_latency[] = {
#include "igemm_tuned_latency.inc"
    {-1, 0, 0, 0, 0, 0, 0, 0}};
std::atomic<int> g_tuning_profile{0};

// Run-time rows in front of the compiled-in table (A/B runs of tools/tune_concurrent.py without a rebuild): the file named
// by MADM_TUNED_FILE holds one "dtype M N K KH variant tile splitk" row per line ('#' starts a comment); read once.
const std::vector<Tuned>& tuned_overrides() {
    static const std::vector<Tuned> rows = [] {
        std::vector<Tuned> v;
        const char* path = getenv("MADM_TUNED_FILE");
        if (!path || !*path) return v;
        FILE* f = fopen(path, "r");
        if (!f) { fprintf(stderr, "madm: MADM_TUNED_FILE=%s cannot be opened\n", path); return v; }
        char line[256];
        while (fgets(line, sizeof line, f)) {
            Tuned t;
            if (line[0] == '#') continue;
            if (sscanf(line, "%d %d %d %d %d %d %d %d", &t.dtype, &t.M, &t.N, &t.K, &t.KH, &t.variant, &t.tile, &t.splitk) != 8)
                continue;
            // a row with an unknown tile code would fall through to the default igemm launch unnoticed: refuse it loudly
            if (t.tile < 1 || t.tile > 17 || t.splitk < 1 || t.variant < 0 || t.variant > 3 || !madm_dtype_ok(t.dtype)) {
                fprintf(stderr, "madm: MADM_TUNED_FILE=%s: row ignored (tile 1..17, splitk >= 1, variant 0..3): %s", path, line);
                continue;
            }
            v.push_back(t);
        }
        fclose(f);
        return v;
    }();
    return rows;
}

const Tuned* find_tuned(int dtype, int M, int N, int K, int KH, int variant) {
    if (dtype == MADM_F16) dtype = MADM_BF16;   // same kernels, same instruction rate: the bf16 table serves both
    if (g_tile_override != 0) return nullptr;
    for (const Tuned& t : tuned_overrides())
        if (t.dtype == dtype && t.M == M && t.N == N && t.K == K && t.KH == KH && t.variant == variant) return &t;
    if (g_tuning_profile.load(std::memory_order_relaxed) == 1)
        for (const Tuned* t = g_tuned_latency; t->dtype >= 0; ++t)
            if (t->dtype == dtype && t->M == M && t->N == N && t->K == K && t->KH == KH && t->variant == variant) return t;
    for (const Tuned* t = g_tuned; t->dtype >= 0; ++t)
        if (t->dtype == dtype && t->M == M && t->N == N && t->K == K && t->KH == KH && t->variant == variant) return t;
    return variant == 3 ? find_tuned(dtype, M, N, K, KH, 0) : nullptr;
}

int heuristic_tile(int M, int N, int K) {
    auto tiles = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    // shapes without a tuned entry (the segmentation head, the training step's gradients, other batch sizes): the
    // register-staged 128 x 128 tile never wins a tuned entry and loses 25 .. 40 % on the head's M = 524 288 GEMMs
    // (M524288: N256 K1024 851 us vs 612 (tile 8) / 630 (tile 2); N1024 K256 1447 vs 892 (tile 2); MI355X, f16)
    if (M >= 128 && tiles(128, 64) >= 256) return K >= 1024 ? 8 : 2;
    return 3;
}

// narrowest map the halo kernels take (env MADM_HALO_MIN_W for A/B runs; ops.can_fuse_groupnorm mirrors it): an 8-wide
// map wastes half of every 8 x 16 patch, but lets the 8 x 8 UNet level fuse its GroupNorm
int halo_min_width() {
    static const int w = [] { const char* e = getenv("MADM_HALO_MIN_W"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : v; }();
    return w;
}

// the LDS halo-tile kernel (conv3x3.hip) handles 3x3 / stride 1 / pad 1 convs on maps of at least one patch
bool halo_eligible(const madm_conv2d_args* a) {
    return a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad_t == 1 && a->pad_l == 1 && !a->upsample &&
           a->OH == a->IH && a->OW == a->IW && a->OH >= 8 && a->OW >= halo_min_width() && a->epilogue != MADM_EPI_GEGLU;
}

// tile codes: 1 = igemm 128x128, 2 = igemm 128x64, 3 = igemm 64x64, 4 = halo conv3x3 BN=128, 5 = halo BN=64,
// 6 = igemm 64x64 with an 8-deep register prefetch (latency-bound small-M GEMMs streaming cold weights),
// 7 / 8 = LDS-DMA igemm 64x64 (4-slot ring) / 128x64 (3-slot ring), 9 / 10 = halo BN=128 / BN=64 with LDS-DMA weights,
// 11 = LDS-DMA igemm 64x64 with a 3-slot ring (48 KB + the 3 KB constant stash of every igemm_glds instantiation = 51 KB:
//      three blocks per CU use 153 of the 160 KB, for grids of 513 .. 768 tiles),
// 12 = halo conv3x3 on 16 x 16-pixel patches, BN = 128, halo and weights by LDS-DMA (conv3x3_h16.hip; maps >= 16 x 16),
// 14 / 15 = LDS-DMA igemm 128x128 with a 3-slot (96 + 3 KB, one block per CU) / 2-slot ring (64 + 3 KB, two blocks per CU): 8 KB of
// operands per MFLOP through the CU's load path instead of 11 (128x64) / 31 (64x64) -- few, fat workgroups for launches
// whose neighbours on the chip are other streams' kernels (tools/tune_concurrent.py)
int pick_tile_raw(const madm_conv2d_args* a) {
    const int M = a->B * a->OH * a->OW, K = a->KH * a->KW * (a->C1 + a->C2);
    const bool halo_ok = halo_eligible(a);
    const int halo_default = (a->N % 128 == 0 || a->N >= 512) ? 4 : 5;
    if (a->gn_sums1) {   // fused GroupNorm exists only in the halo kernels
        if (is_halo_tile(g_tile_override)) return g_tile_override;
        if (const Tuned* t = find_tuned(a->dtype, M, a->N, K, a->KH, variant_of(a)))
            if (is_halo_tile(t->tile)) return t->tile;
        return halo_default;
    }
    if (g_tile_override > 0 && (is_igemm_tile(g_tile_override) || (is_halo_tile(g_tile_override) && halo_ok)))
        return g_tile_override;   // (13 = the A-stationary kernel is handled by pick_tile; ineligible launches fall through)
    if (const Tuned* t = find_tuned(a->dtype, M, a->N, K, a->KH, variant_of(a)))
        if (is_igemm_tile(t->tile) || halo_ok) return t->tile;
    if (halo_ok && M >= 2048) return halo_default;
    return heuristic_tile(M, a->N, K);
}

// the 16 x 16-patch kernel pays where it fills the chip: at least ~0.75 rounds of its 256-pixel x 128-channel blocks
// (measured against tile 9 on MI355X, bf16 / f16: +12 .. 23 % on the 512^2 .. 128^2 maps of the VAE, 0.6 x on an 8192-pixel map)
bool h16_pays(const madm_conv2d_args* a) {
    static const int off = [] { const char* e = getenv("MADM_NO_H16"); return e ? atoi(e) : 0; }();
    if (off || a->OH < 16 || a->OW < 16 || a->N < 128) return false;
    const long long blocks = (long long)a->B * ((a->OH + 15) / 16) * ((a->OW + 15) / 16) * ((a->N + 127) / 128);
    return blocks >= 384;
}

// nearest-2x upsample + 3x3 conv (Upsample2D of the VAE decoder / UNet): only the 16 x 16-patch kernel folds the
// upsample into its halo gather (igemm 128x64 on the 256-channel 512 x 512 layer: 456 us, this kernel: see DESIGN.md)
bool h16_upsample_eligible(const madm_conv2d_args* a) {
    return a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad_t == 1 && a->pad_l == 1 && a->upsample &&
           a->OH == 2 * a->IH && a->OW == 2 * a->IW && !a->gn_sums1 && a->epilogue != MADM_EPI_GEGLU && a->OH >= 16 &&
           a->OW >= 16;
}

int pick_tile(const madm_conv2d_args* a) {
    if (apanel_eligible(a)) {
        if (g_tile_override == 13) return 13;
        if (g_tile_override == 0) {
            const int M = a->B * a->OH * a->OW;
            if (const Tuned* t = find_tuned(a->dtype, M, a->N, a->C1, 1, 0))
                if (t->tile == 13) return 13;
        }
    }
    if (h16_upsample_eligible(a)) {
        if (g_tile_override == 12) return 12;
        if (g_tile_override == 0) {
            // a table row decides (variant 2; the side-by-side tuner put the UNet's upsample convs here although their grids
            // are far below a round of workgroups), h16_pays() where there is none
            const int M = a->B * a->OH * a->OW, K = a->KH * a->KW * (a->C1 + a->C2);
            if (const Tuned* t = find_tuned(a->dtype, M, a->N, K, a->KH, 2)) {
                if (t->tile == 12) return 12;
            } else if (h16_pays(a)) {
                return 12;
            }
        }
    }
    const int t = pick_tile_raw(a);
    if (t == 12 && (a->OH < 16 || a->OW < 16)) return 9;   // the 16 x 16-patch kernel needs a map of at least one patch
    if ((t == 4 || t == 9) && g_tile_override == 0 && h16_pays(a)) return 12;
    return t;
}

void tile_dims(int t, int& bm, int& bn) {
    if (t == 12) { bm = 256; bn = 128; }
    else if (t == 1 || t == 4 || t == 9 || t == 14 || t == 15) { bm = 128; bn = 128; }
    else if (t == 2 || t == 5 || t == 8 || t == 10 || t == 17) { bm = 128; bn = 64; }
    else { bm = 64; bn = 64; }
}

int fill_params(const madm_conv2d_args* a, IgemmP& p) {
    MADM_REQUIRE(a != nullptr, "conv2d: null args");
    MADM_REQUIRE(madm_dtype_ok(a->dtype), "conv2d: bad dtype %d", a->dtype);
    const int bke = (8 * madm_epc(a->dtype));
    MADM_REQUIRE(a->in1 && a->w && a->out, "conv2d: null tensor pointer");
    MADM_REQUIRE(a->C1 > 0 && a->C1 % bke == 0, "conv2d: C1=%d must be a positive multiple of %d", a->C1, bke);
    MADM_REQUIRE(a->C2 >= 0 && a->C2 % bke == 0, "conv2d: C2=%d must be a multiple of %d", a->C2, bke);
    MADM_REQUIRE(a->C2 == 0 || a->in2, "conv2d: C2>0 needs in2");
    MADM_REQUIRE(a->B > 0 && a->IH > 0 && a->IW > 0 && a->OH > 0 && a->OW > 0, "conv2d: bad dims");
    MADM_REQUIRE(a->KH > 0 && a->KW > 0 && a->stride > 0, "conv2d: bad kernel/stride");
    MADM_REQUIRE(a->N > 0 && a->N % 4 == 0, "conv2d: N=%d must be a positive multiple of 4", a->N);
    MADM_REQUIRE(a->epilogue >= MADM_EPI_NONE && a->epilogue <= MADM_EPI_RELU, "conv2d: bad epilogue");
    MADM_REQUIRE(a->splitk >= 1, "conv2d: splitk must be >= 1");
    const int ocols = (a->epilogue == MADM_EPI_GEGLU) ? a->N / 2 : a->N;
    MADM_REQUIRE(a->ldo >= ocols && a->ldo % 2 == 0, "conv2d: ldo=%d too small/odd for %d columns", a->ldo, ocols);
    MADM_REQUIRE(a->epilogue == MADM_EPI_GEGLU || a->ldo % 4 == 0, "conv2d: ldo must be a multiple of 4");
    MADM_REQUIRE(!a->residual || (a->ldr >= ocols && a->ldr % 2 == 0), "conv2d: bad ldr");
    p.in1 = (const char*)a->in1; p.in2 = (const char*)a->in2; p.w = (const char*)a->w;
    p.bias = a->bias; p.rowvec = a->rowvec; p.residual = (const char*)a->residual;
    p.out = (char*)a->out; p.ws = (float*)a->workspace; p.stats = a->stats;
    p.gn_sums1 = nullptr; p.gn_sums2 = nullptr; p.gn_gamma = nullptr; p.gn_beta = nullptr;
    p.gn_G = 0; p.gn_eps = 0.f; p.gn_magic = 0; p.act = 0;
    p.ln_cs = a->ln_colsum; p.ln_eps = a->ln_eps;
    if (a->ln_colsum) {
        MADM_REQUIRE(a->KH == 1 && a->KW == 1 && a->stride == 1 && a->pad_t == 0 && a->pad_l == 0 && !a->upsample &&
                     a->C2 == 0 && a->OH == a->IH && a->OW == a->IW,
                     "conv2d: the folded LayerNorm needs a linear layer / 1x1 conv over ONE source (K = C1)");
        MADM_REQUIRE(!a->gn_sums1 && a->ln_eps > 0.f && a->splitk == 1,
                     "conv2d: folded LayerNorm: no fused GroupNorm, eps > 0, splitk == 1 (every block must see whole rows)");
    }
    MADM_REQUIRE(!a->stats || a->epilogue != MADM_EPI_GEGLU, "conv2d: fused statistics cannot follow GEGLU");
    p.C1 = a->C1; p.C2 = a->C2; p.Ctot = a->C1 + a->C2;
    p.B = a->B; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW;
    p.KH = a->KH; p.KW = a->KW; p.stride = a->stride; p.pad_t = a->pad_t; p.pad_l = a->pad_l;
    p.upsample = a->upsample ? 1 : 0;
    p.N = a->N; p.K = a->KH * a->KW * p.Ctot; p.M = a->B * a->OH * a->OW;
    MADM_REQUIRE(!a->rowvec || (a->ldrv >= a->N && a->ldrv % 4 == 0), "conv2d: bad ldrv=%d", a->ldrv);
    p.ldr = a->ldr; p.ldo = a->ldo; p.ldrv = a->ldrv; p.epilogue = a->epilogue;
    p.ldw = a->ldw ? a->ldw : p.K;
    p.out_f32 = a->out_f32 ? 1 : 0;
    MADM_REQUIRE(p.ldw >= p.K && p.ldw % (bke / 8) == 0, "conv2d: bad weight row stride ldw=%d", p.ldw);
    MADM_REQUIRE(!p.out_f32 || (a->epilogue != MADM_EPI_GEGLU && !a->residual), "conv2d: out_f32 cannot follow GEGLU / residual");
    p.ld1 = a->ld1 ? a->ld1 : a->C1;
    p.ld2 = a->ld2 ? a->ld2 : a->C2;
    MADM_REQUIRE(p.ld1 >= a->C1 && p.ld2 >= a->C2 && p.ld1 % (bke / 8) == 0 && p.ld2 % (bke / 8) == 0,
                 "conv2d: bad source row strides ld1=%d ld2=%d", p.ld1, p.ld2);
    {
        const size_t es = madm_esize(a->dtype);
        const size_t px = (size_t)a->B * a->IH * a->IW;
        const size_t b1 = ((px - 1) * p.ld1 + a->C1) * es;
        const size_t b2 = a->C2 ? ((px - 1) * p.ld2 + a->C2) * es : 0;
        const size_t bw = ((size_t)(a->N - 1) * p.ldw + p.K) * es;
        MADM_REQUIRE(b1 < 0x80000000ull && b2 < 0x80000000ull && bw < 0x80000000ull,
                     "conv2d: tensors must stay below 2 GiB (32-bit buffer offsets)");
        p.bytes1 = (unsigned)b1; p.bytes2 = (unsigned)b2; p.bytesw = (unsigned)bw;
    }
    p.nk = p.K / bke;
    p.splitk = a->splitk > p.nk ? p.nk : a->splitk;
    if (p.splitk < 1) p.splitk = 1;
    MADM_REQUIRE((long long)p.nk * (p.splitk + 1) < 0x7fffffffLL, "conv2d: K too large for the 32-bit slice arithmetic");
    return MADM_OK;
}

template <typename T, int BM, int BN, int NS, bool LIN>
int launch_glds_v(const IgemmP& p, dim3 grid, hipStream_t s) {
    constexpr size_t lds = (size_t)NS * (BM + BN) * 128;
    auto kern = igemm_glds_kernel<T, BM, BN, NS, LIN>;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = madm_raise_dynamic_lds(reinterpret_cast<const void*>(kern), (size_t)(lds), attr_done, "igemm (LDS-DMA)")) return e;
    kern<<<grid, 256, lds, s>>>(p);
    return madm_check_launch("igemm_glds_kernel");
}

template <typename T, int BM, int BN, int NS>
int launch_glds(const IgemmP& p, dim3 grid, hipStream_t s) {
    const bool lin = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && !p.upsample &&
                     p.OH == p.IH && p.OW == p.IW;
    return lin ? launch_glds_v<T, BM, BN, NS, true>(p, grid, s) : launch_glds_v<T, BM, BN, NS, false>(p, grid, s);
}

template <typename T>
int launch(const IgemmP& p0, int t, hipStream_t s, const PostGn& pn) {
    IgemmP p = p0;
    int bm, bn;
    tile_dims(t, bm, bn);
    int rc;
    if (t == 13) return launch_igemm_apanel<T>(p, s);
    if (is_halo_tile(t)) {
        rc = (t == 12) ? launch_conv3x3_h16<T>(p, bn, s)
                       : ((t >= 9) ? launch_conv3x3_halo_dma<T>(p, bn, s) : launch_conv3x3_halo<T>(p, bn, s));
    } else {
        p.tilesN = (p.N + bn - 1) / bn;
        const int tilesM = (p.M + bm - 1) / bm;
        dim3 grid((unsigned)(tilesM * p.tilesN), 1, (unsigned)p.splitk);
        const bool lin = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && !p.upsample &&
                         p.OH == p.IH && p.OW == p.IW;
        if (t == 1) igemm_kernel<T, 128, 128, 2, false><<<grid, 256, 0, s>>>(p);
        else if (t == 2 && lin) igemm_kernel<T, 128, 64, 3, true><<<grid, 256, 0, s>>>(p);
        else if (t == 2) igemm_kernel<T, 128, 64, 3, false><<<grid, 256, 0, s>>>(p);
        else if (t == 6) igemm_kernel<T, 64, 64, 8, false><<<grid, 256, 0, s>>>(p);
        else if (t == 7) { if (int e = launch_glds<T, 64, 64, 4>(p, grid, s)) return e; }
        else if (t == 8) { if (int e = launch_glds<T, 128, 64, 3>(p, grid, s)) return e; }
        else if (t == 11) { if (int e = launch_glds<T, 64, 64, 3>(p, grid, s)) return e; }
        else if (t == 14) { if (int e = launch_glds<T, 128, 128, 3>(p, grid, s)) return e; }
        else if (t == 15) { if (int e = launch_glds<T, 128, 128, 2>(p, grid, s)) return e; }
        // two-slot rings of the small tiles: 32 / 48 KB rings + the 3 KB constant stash = 35 / 51 KB of LDS: four / three blocks
        // per CU by LDS (short-K layers: the blocks' prologues and epilogues cover each other instead of a deep ring covering
        // the K loop)
        else if (t == 16) { if (int e = launch_glds<T, 64, 64, 2>(p, grid, s)) return e; }
        else if (t == 17) { if (int e = launch_glds<T, 128, 64, 2>(p, grid, s)) return e; }
        else if (lin) igemm_kernel<T, 64, 64, 4, true><<<grid, 256, 0, s>>>(p);
        else igemm_kernel<T, 64, 64, 4, false><<<grid, 256, 0, s>>>(p);
        rc = madm_check_launch("igemm_kernel");
    }
    if (rc) return rc;
    if (p.splitk > 1 && pn.gamma) {
        const size_t lds = post_gn_lds(p.OH * p.OW, p.N, pn.G);
        const dim3 pgrid((unsigned)pn.G, (unsigned)p.B);
        if ((p.N / pn.G) % 4 == 0) {
            auto kern = splitk_groupnorm_kernel<T, 4>;
            static std::atomic<uint64_t> attr_done{0};
            // the attribute is raised ONCE per device, so to the kernel's maximum, not to this launch's size (a later, larger
            // group -- another resolution, train then eval in one process -- must still launch)
            if (int e = madm_raise_dynamic_lds(reinterpret_cast<const void*>(kern), PGN_MAX_LDS, attr_done, "split-K + GroupNorm")) return e;
            kern<<<pgrid, PGN_THREADS, lds, s>>>(p, pn);
        } else {
            auto kern = splitk_groupnorm_kernel<T, 2>;
            static std::atomic<uint64_t> attr_done{0};
            if (int e = madm_raise_dynamic_lds(reinterpret_cast<const void*>(kern), PGN_MAX_LDS, attr_done, "split-K + GroupNorm")) return e;
            kern<<<pgrid, PGN_THREADS, lds, s>>>(p, pn);
        }
        rc = madm_check_launch("splitk_groupnorm_kernel");
    } else if (p.splitk > 1) {
        dim3 rgrid((unsigned)((p.N / 4 + 15) / 16), (unsigned)((p.M + 15) / 16));
        splitk_reduce_kernel<T><<<rgrid, 256, 0, s>>>(p);
        rc = madm_check_launch("splitk_reduce_kernel");
    }
    return rc;
}

}  // namespace

extern "C" {

#ifdef GLDS_STAMPS
int madm_debug_read_reg_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_reg_stamps), sizeof(unsigned long long) * n);
}
int madm_debug_read_glds_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_glds_stamps), sizeof(unsigned long long) * n);
}
#endif

void madm_debug_set_conv_tile(int t) { g_tile_override = t; }

int madm_set_tuning_profile(int profile) {
    MADM_REQUIRE(profile == 0 || profile == 1, "set_tuning_profile: 0 = throughput (side-by-side rows), 1 = latency (lone-launch rows)");
    g_tuning_profile.store(profile, std::memory_order_relaxed);
    return MADM_OK;
}
int madm_get_tuning_profile(void) { return g_tuning_profile.load(std::memory_order_relaxed); }

size_t madm_conv2d_workspace_bytes(const madm_conv2d_args* a) {
    if (!a || a->splitk <= 1) return 0;
    const size_t M = (size_t)a->B * a->OH * a->OW;
    return (size_t)a->splitk * M * (size_t)a->N * sizeof(float);
}

int madm_conv2d_pick_tile(const madm_conv2d_args* a) {
    if (!a) return 0;
    return pick_tile(a);
}

int madm_conv2d_has_tuned_row(const madm_conv2d_args* a) {
    if (!a || !madm_dtype_ok(a->dtype)) return 0;
    const int M = a->B * a->OH * a->OW, K = a->KH * a->KW * (a->C1 + a->C2);
    if (h16_upsample_eligible(a) && find_tuned(a->dtype, M, a->N, K, a->KH, 2)) return 1;
    return find_tuned(a->dtype, M, a->N, K, a->KH, variant_of(a)) ? 1 : 0;
}

int madm_conv2d_suggest_splitk(const madm_conv2d_args* a) {
    if (!a) return 1;
    const int bke = (8 * madm_epc(a->dtype));
    const int M = a->B * a->OH * a->OW;
    const int Ktot = a->KH * a->KW * (a->C1 + a->C2);
    const int nk = Ktot / bke;
    const int chosen = pick_tile(a);
    if (const Tuned* t = find_tuned(a->dtype, M, a->N, Ktot, a->KH, variant_of(a)))
        if (t->tile == chosen) return t->splitk;
    int bm, bn;
    tile_dims(pick_tile(a), bm, bn);
    const long long tiles = (long long)((M + bm - 1) / bm) * ((a->N + bn - 1) / bn);
    if (tiles >= 192 || nk < 8) return 1;
    long long s = (512 + tiles - 1) / tiles;
    if (s > nk / 4) s = nk / 4;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    return (int)s;
}

int madm_conv2d_fwd(const madm_conv2d_args* a, void* stream) {
    IgemmP p;
    int rc = fill_params(a, p);
    if (rc) return rc;
    if (p.splitk > 1) {
        const size_t need = (size_t)p.splitk * p.M * (size_t)p.N * sizeof(float);
        MADM_REQUIRE(a->workspace && a->workspace_bytes >= need,
                     "conv2d: split-K %d needs %zu workspace bytes, got %zu", p.splitk, need,
                     a->workspace_bytes);
    }
    const int t = pick_tile(a);
    if (a->gn_sums1) {
        MADM_REQUIRE(halo_eligible(a),
                     "conv2d: fused GroupNorm needs a 3x3 / stride-1 / pad-1 conv on a map of at least 8x16 "
                     "(use madm_groupnorm_apply otherwise; madm_conv2d_can_fuse_groupnorm tells)");
        MADM_REQUIRE(a->gn_gamma && a->gn_beta && (a->C2 == 0 || a->gn_sums2), "conv2d: fused GroupNorm needs gamma, beta "
                     "and the sums of every source");
        MADM_REQUIRE(a->gn_groups > 0 && a->gn_groups <= 32 && p.Ctot % a->gn_groups == 0 && a->gn_eps > 0.f,
                     "conv2d: fused GroupNorm: bad groups=%d (C=%d) / eps", a->gn_groups, p.Ctot);
        MADM_REQUIRE(a->gn_act >= 0 && a->gn_act <= 2, "conv2d: bad gn_act %d", a->gn_act);
        const unsigned cpg = (unsigned)(p.Ctot / a->gn_groups);
        const unsigned magic = (unsigned)((0x100000000ull + cpg - 1) / cpg);   // c / cpg == umulhi(c, magic), checked:
        for (unsigned c = 0; c < (unsigned)p.Ctot; ++c)
            MADM_REQUIRE((unsigned)(((unsigned long long)c * magic) >> 32) == c / cpg, "conv2d: group index magic failed");
        p.gn_sums1 = a->gn_sums1; p.gn_sums2 = a->gn_sums2; p.gn_gamma = a->gn_gamma; p.gn_beta = a->gn_beta;
        p.gn_G = a->gn_groups; p.gn_eps = a->gn_eps; p.gn_magic = magic; p.act = a->gn_act;
    }
    if (is_halo_tile(t)) {   // the halo kernel splits K by whole channel chunks
        const int nchunks = p.Ctot / ((8 * madm_epc(a->dtype)));
        if (p.splitk > nchunks) p.splitk = nchunks;
    }
    PostGn pn{nullptr, nullptr, 0, 0.f, 0};
    if (a->pn_gamma) {
        MADM_REQUIRE(madm_conv2d_can_post_groupnorm(a),
                     "conv2d: the GroupNorm of the output rides on the split-K reduction: effective splitk > 1, plain "
                     "epilogue, no residual / stats / out_f32, even N / groups, HW * N / groups * 4 <= %zu bytes of LDS "
                     "(madm_conv2d_can_post_groupnorm tells)", PGN_MAX_LDS);
        MADM_REQUIRE(a->pn_beta && a->pn_eps > 0.f && a->pn_act >= 0 && a->pn_act <= 2, "conv2d: bad pn_beta / pn_eps / pn_act");
        pn = PostGn{a->pn_gamma, a->pn_beta, a->pn_groups, a->pn_eps, a->pn_act};
    }
    hipStream_t s = (hipStream_t)stream;
    if (a->dtype == MADM_F32) return launch<float>(p, t, s, pn);
    if (a->dtype == MADM_F16) return launch<f16_t>(p, t, s, pn);
    return launch<bf16_t>(p, t, s, pn);
}

int madm_conv2d_can_post_groupnorm(const madm_conv2d_args* a) {
    if (!a || !madm_dtype_ok(a->dtype) || a->splitk <= 1 || a->pn_groups <= 0 || a->N <= 0 || a->N % a->pn_groups) return 0;
    if (a->epilogue != MADM_EPI_NONE || a->residual || a->stats || a->out_f32 || a->ln_colsum) return 0;
    const int bke = 8 * madm_epc(a->dtype);
    const int Ctot = a->C1 + a->C2;
    int sk = a->splitk;                                   // the clamps of fill_params / madm_conv2d_fwd
    const int nk = a->KH * a->KW * Ctot / bke;
    if (sk > nk) sk = nk;
    if (is_halo_tile(pick_tile(a)) && sk > Ctot / bke) sk = Ctot / bke;
    if (sk <= 1) return 0;
    const int cpg = a->N / a->pn_groups;
    return (cpg % 2 == 0 && post_gn_lds(a->OH * a->OW, a->N, a->pn_groups) <= PGN_MAX_LDS) ? 1 : 0;
}

int madm_conv2d_can_fuse_groupnorm(const madm_conv2d_args* a) { return a && halo_eligible(a) ? 1 : 0; }

}  // extern "C"
