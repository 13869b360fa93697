// Backward of madm_conv2d_fwd with respect to the weights, and the weight repack its data gradient needs.
//   dw[n][k] += sum_m dout[m][n] * A(m, k)       k = (kh, kw, c), A = the gather of igemm.hip
// Both operands of this GEMM are contiguous along the OUTPUT dims (n resp. c) and strided along the reduction dim (the
// pixel m), i.e. both are needed transposed.  The tiles are staged into LDS as they lie in memory ([pixel][channel]
// rows) and the MFMA fragments are read transposed: bf16 with ds_read_b64_tr_b16 (a 4-pixel x 16-channel block
// transposed by the LDS: lane (fi, fg) receives pixels 4 fg .. 4 fg + 3 of channel fi), f32 with four ds_read_b32.
// Lane group fg therefore holds pixels {4 fg + r, 16 + 4 fg + r} of a 32-pixel step in its eight k slots -- the same
// permutation of the reduction index for both operands, so the sum is unchanged.
// Block = 128 output channels x 128 weight columns, 4 waves 2 x 2, K loop over pixels in steps of 32 (bf16) / 16 (f32)
// with a register-prefetch double buffer, grid.z slices the pixel range (the weight gradient of a 512 x 512 layer
// reduces over 524 288 pixels into a 128 x 1152 tile: without slicing it would be nine blocks); partial tiles are
// added into the f32 gradient with hardware float atomics -- which also gives the "+=" of gradient accumulation.
// The data gradient of a stride-1 conv / linear layer is madm_conv2d_fwd itself on dout with the weights transposed
// and the taps reversed: madm_pack_dgrad_weights produces that layout.
// The reference gets both from torch autograd (loss.backward(), engine/train_loop.py:203-217).
#include "common.hpp"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

struct WgradP {
    const char* in1; const char* in2; const char* dout; float* dw;
    int C1, C2, Ctot, ld1, ld2, ldd, B, IH, IW, OH, OW, KH, KW, stride, pad_t, pad_l, upsample;
    int N, K, M, steps_per_slice, lin;
    float* dbias;
    unsigned bytes1, bytes2, bytesd;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv2d_wgrad_kernel(const WgradP p) {
    constexpr int EPC = TT<T>::EPC;
    constexpr int PX = 4 * EPC;                  // pixels per K step: one MFMA k-extent (bf16 32, f32 16)
    constexpr int CPR = 128 / EPC;               // 16-byte chunks per tile row
    constexpr int RPP = 256 / CPR;               // rows staged per pass of the block
    constexpr int LI = PX / RPP;                 // passes (2)
    constexpr int ROWB = 128 * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);   // padded: transposed reads conflict-free
    constexpr int TILEB = PX * ROWB;
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) char smem[4 * TILEB];   // {dout, A} x 2 stages

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wc = wave >> 1;
    const int fi = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 128, kc0 = blockIdx.y * 128;
    const int cc = tid % CPR, r0 = tid / CPR;

    const int step0 = blockIdx.z * p.steps_per_slice;
    const int total_steps = (p.M + PX - 1) / PX;
    const int nsteps = min(p.steps_per_slice, total_steps - step0);
    if (nsteps <= 0) return;
    const int mend = (step0 + nsteps) * PX < p.M ? (step0 + nsteps) * PX : p.M;   // first pixel beyond this slice

    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dout, 0, p.bytesd, 0x00020000);
    // this thread's weight column: tap and channel are fixed, only the pixel moves
    const int kc = kc0 + cc * EPC;
    const bool kc_ok = kc < p.K;
    const int tap = kc_ok ? kc / p.Ctot : 0;
    const int c = kc - tap * p.Ctot;
    const int tr = tap / p.KW, ts = tap - tr * p.KW;
    const bool first = c < p.C1;
    const __amdgpu_buffer_rsrc_t rsa =
        first ? __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.bytes1, 0x00020000)
              : __builtin_amdgcn_make_buffer_rsrc((void*)p.in2, 0, p.bytes2, 0x00020000);
    const int lda = first ? p.ld1 : p.ld2;
    const int ca = first ? c : c - p.C1;
    const bool n_ok = n0 + cc * EPC < p.N;

    // output pixel of staged row i: m = (step0 + s) * PX + r0 + RPP * i, kept as (image, y, x) and advanced by PX
    int pm[LI], pb[LI], py[LI], px[LI];
#pragma unroll
    for (int i = 0; i < LI; ++i) {
        pm[i] = step0 * PX + r0 + RPP * i;
        const int q = pm[i] / p.OW;
        px[i] = pm[i] - q * p.OW;
        pb[i] = q / p.OH;
        py[i] = q - pb[i] * p.OH;
    }
    const int IHe = p.upsample ? 2 * p.IH : p.IH, IWe = p.upsample ? 2 * p.IW : p.IW;

    // Two register sets: the operands of step s + 2 are requested before step s is multiplied and written to LDS after
    // step s + 1 -- two MFMA phases of lead instead of none (one set: every step waited a full operand round trip, ~1 us
    // for 256 clocks of MFMA work: tools/exp/sweep_wgrad_splitm.py, one slice).  No condition around a request (steps past
    // the slice read out-of-range offsets = zeros, and are multiplied as zeros): the compiler keeps counted waits only in
    // straight-line code; three rotating sets under conditions made it drain every step.
    u32x4 rdA[LI], raA[LI], rdB[LI], raB[LI];
    auto load_step = [&](u32x4 (&rd)[LI], u32x4 (&ra)[LI]) {
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            // branch-free per lane: an invalid row / column / tap position sets bit 31 of the offset (out of range for the
            // descriptor: zeros).  Written as "cond ? off : OOB" the compiler built two loads behind exec masks and had to
            // drain vmcnt between them (the address temporary shared the destination's registers).
            const bool m_ok = pm[i] < mend;   // (steps past the slice must not read the next slice's pixels)
            const unsigned badd = (unsigned)(!(m_ok && n_ok)) << 31;
            const unsigned offd = ((unsigned)(pm[i] * p.ldd + n0 + cc * EPC) * (unsigned)sizeof(T)) | badd;
            rd[i] = __builtin_amdgcn_raw_buffer_load_b128(rsd, offd, 0, 0);
            unsigned offa;
            bool a_ok = m_ok && kc_ok;
            if (p.lin) {   // uniform
                offa = (unsigned)(pm[i] * lda + ca) * (unsigned)sizeof(T);
            } else {
                int iy = py[i] * p.stride - p.pad_t + tr, ix = px[i] * p.stride - p.pad_l + ts;
                a_ok = a_ok && (unsigned)iy < (unsigned)IHe && (unsigned)ix < (unsigned)IWe;
                if (p.upsample) { iy >>= 1; ix >>= 1; }   // uniform
                offa = (unsigned)(((pb[i] * p.IH + iy) * p.IW + ix) * lda + ca) * (unsigned)sizeof(T);
            }
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsa, offa | ((unsigned)(!a_ok) << 31), 0, 0);
            pm[i] += PX;
            if (!p.lin) {   // (a linear layer is OW = 1: the walk below would take PX iterations per step for nothing)
                px[i] += PX;
                while (px[i] >= p.OW) {
                    px[i] -= p.OW;
                    if (++py[i] == p.OH) { py[i] = 0; ++pb[i]; }
                }
            }
        }
    };
    // bias gradient: the blocks of the first weight-column tile add up the dout chunks they stage (rows beyond M arrive
    // as zeros); reduced over the block's row slots at the end
    const bool do_bias = p.dbias != nullptr && blockIdx.y == 0;
    float bsum[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) bsum[j] = 0.f;
    auto store_step = [&](int stage, const u32x4 (&rd)[LI], const u32x4 (&ra)[LI]) {
        char* d = smem + stage * 2 * TILEB;
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            const int off = (r0 + RPP * i) * ROWB + cc * 16;
            *reinterpret_cast<u32x4*>(d + off) = rd[i];
            *reinterpret_cast<u32x4*>(d + TILEB + off) = ra[i];
            if (do_bias) {
                float f[EPC];
                chunk_to_f32<T>(__builtin_bit_cast(uint4, rd[i]), f);
#pragma unroll
                for (int j = 0; j < EPC; ++j) bsum[j] += f[j];
            }
        }
    };

    f32x4 acc[4][4];   // [jn: 16 output channels][ic: 16 weight columns]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_step(rdA, raA);   // step 0
    load_step(rdB, raB);   // step 1
    store_step(0, rdA, raA);
    __syncthreads();
    auto compute_step = [&](int stage) {
        const char* td = smem + stage * 2 * TILEB;
        const char* ta = td + TILEB;
        uint4 fd[4], fa[4];
        if constexpr (sizeof(T) == 2) {
            const int row = fg * 4 + (fi >> 2), colb = (fi & 3) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* q = td + row * ROWB + (wn * 64 + j * 16) * 2 + colb;
                const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)q);
                const s16x4 x1 =
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + 16 * ROWB));
                const uint2 a0 = __builtin_bit_cast(uint2, x0), a1 = __builtin_bit_cast(uint2, x1);
                fd[j] = make_uint4(a0.x, a0.y, a1.x, a1.y);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const char* q = ta + row * ROWB + (wc * 64 + i * 16) * 2 + colb;
                const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)q);
                const s16x4 x1 =
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + 16 * ROWB));
                const uint2 a0 = __builtin_bit_cast(uint2, x0), a1 = __builtin_bit_cast(uint2, x1);
                fa[i] = make_uint4(a0.x, a0.y, a1.x, a1.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* q = td + (fg * 4) * ROWB + (wn * 64 + j * 16 + fi) * 4;
                float4 v;
                v.x = *reinterpret_cast<const float*>(q);
                v.y = *reinterpret_cast<const float*>(q + ROWB);
                v.z = *reinterpret_cast<const float*>(q + 2 * ROWB);
                v.w = *reinterpret_cast<const float*>(q + 3 * ROWB);
                fd[j] = __builtin_bit_cast(uint4, v);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const char* q = ta + (fg * 4) * ROWB + (wc * 64 + i * 16 + fi) * 4;
                float4 v;
                v.x = *reinterpret_cast<const float*>(q);
                v.y = *reinterpret_cast<const float*>(q + ROWB);
                v.z = *reinterpret_cast<const float*>(q + 2 * ROWB);
                v.w = *reinterpret_cast<const float*>(q + 3 * ROWB);
                fa[i] = __builtin_bit_cast(uint4, v);
            }
        }
        // D[i][j] of mma16: i = row of the first operand (4 per lane), j = row of the second (lane & 15): the lane's
        // column index runs along the contiguous dim of dw
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) mma16<T>(fd[j], fa[i], acc[j][i]);
    };
    for (int s = 0; s < nsteps; s += 2) {
        load_step(rdA, raA);            // step s + 2
        compute_step(0);                // step s
        store_step(1, rdB, raB);        // step s + 1
        __syncthreads();
        load_step(rdB, raB);            // step s + 3
        compute_step(1);                // step s + 1 (zeros past the slice)
        store_step(0, rdA, raA);        // step s + 2
        __syncthreads();
    }

    if (do_bias) {   // (the loop's last barrier has passed: the staging LDS is free)
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int j = 0; j < EPC; ++j) red[r0 * 128 + cc * EPC + j] = bsum[j];
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            float t = 0.f;
            for (int r = 0; r < RPP; ++r) t += red[r * 128 + tid];
            unsafeAtomicAdd(p.dbias + n0 + tid, t);
        }
    }

    // lane holds dw[n = nb + 4 fg + r][k = kb + fi]: one instruction covers 4 rows x 64 contiguous bytes.  A single
    // slice owns its tile of dw: plain read-modify-write (L2 float atomics run at ~0.35 T elements/s, a third of that)
    const bool owner = gridDim.z == 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + fg * 4;
        if (owner) {
            float old[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kc0 + wc * 64 + i * 16 + fi;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    old[i][r] = (k < p.K && n + r < p.N) ? p.dw[(size_t)(n + r) * p.K + k] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kc0 + wc * 64 + i * 16 + fi;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k < p.K && n + r < p.N) p.dw[(size_t)(n + r) * p.K + k] = old[i][r] + acc[j][i][r];
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kc0 + wc * 64 + i * 16 + fi;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.N) unsafeAtomicAdd(p.dw + (size_t)(n + r) * p.K + k, acc[j][i][r]);
        }
    }
}

// wt[c][T - 1 - t][n] = w[n][t][c]: 32 x 32 tiles through LDS, coalesced on both sides
template <typename T>
__global__ __launch_bounds__(256) void pack_dgrad_weights_kernel(const T* __restrict__ w, T* __restrict__ wt, int N,
                                                                 int Tn, int C) {
    __shared__ T tile[32][33];
    const int t = blockIdx.z;
    const int n0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, c = c0 + tx;
        if (n < N && c < C) tile[ty + 8 * i][tx] = w[((size_t)n * Tn + t) * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, n = n0 + tx;
        if (n < N && c < C) wt[((size_t)c * Tn + (Tn - 1 - t)) * N + n] = tile[tx][ty + 8 * i];
    }
}

// f32 master weights -> the packed forward operand of madm_conv2d_fwd, one launch per tensor (after every optimizer step
// all ~700 weights are re-derived: as torch ops that was cat + permute + copy + cast + copy per tensor, 2 200 launches and
// 12 ms of a 254 ms training step):  out[n][(tap, padded channel)] = w[row(n)][channel][tap], the channels of every
// concatenated source zero-padded to the K tile.  Block = (output row, 256 padded channels of one source): the
// [channel][tap] block of the row is contiguous in w -> staged through LDS, written tap-major.
// interleave: out row r <- w row (r & 1 ? N / 2 : 0) + r / 2 (GEGLU: value_j / gate_j rows side by side).
struct PackP {
    const float* w; void* out;
    int N, Cin, taps, nsrc, ldo, cpad_tot, interleave;
    int C[4], coff[4], cpad[4], cpoff[4];
};

template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const PackP p) {
    __shared__ float st[256 * 9];
    const int n = blockIdx.x;
    int y = blockIdx.y, s = 0;
    while (s < p.nsrc - 1 && y >= (p.cpad[s] + 255) / 256) { y -= (p.cpad[s] + 255) / 256; ++s; }
    const int cp0 = y * 256;
    const int row = p.interleave ? ((n & 1) ? p.N / 2 : 0) + (n >> 1) : n;
    int cvalid = p.C[s] - cp0;                        // real channels in this block (the rest is padding)
    cvalid = cvalid < 0 ? 0 : (cvalid > 256 ? 256 : cvalid);
    const float* src = p.w + ((size_t)row * p.Cin + p.coff[s] + cp0) * p.taps;
    for (int i = threadIdx.x; i < cvalid * p.taps; i += 256) st[i] = src[i];
    __syncthreads();
    int cw = p.cpad[s] - cp0;
    cw = cw > 256 ? 256 : cw;
    T* orow = reinterpret_cast<T*>(p.out) + (size_t)n * p.ldo + p.cpoff[s] + cp0;
    for (int i = threadIdx.x; i < cw * p.taps; i += 256) {
        const int tap = i / cw, c = i - tap * cw;
        TT<T>::st(orow + (size_t)tap * p.cpad_tot + c, c < cvalid ? st[c * p.taps + tap] : 0.f);
    }
}

// LayerNorm folded into the Linear that consumes it (madm_conv2d_args.ln_colsum), one launch per layer:
//   out[n][k] = T(w[row(n)][k] * gamma[k])   (the f32 product, then rounded to T)
//   bias'[n]  = sum_k w[row(n)][k] * beta[k] + b[row(n)]           (f64 accumulation)
//   colsum[n] = sum_k out[n][k]                                    (f64 sum of the ROUNDED weights: exact)
// one workgroup per output row.
template <typename T>
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             T* __restrict__ out, float* __restrict__ bias_out,
                                                             float* __restrict__ colsum, int N, int K, int interleave) {
    __shared__ double red[2][4];
    const int n = blockIdx.x;
    const int row = interleave ? ((n & 1) ? N / 2 : 0) + (n >> 1) : n;
    const float* src = w + (size_t)row * K;
    double cs = 0.0, bs = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float wv = src[k];
        float pr = wv * gamma[k];
        asm volatile("" : "+v"(pr));     // the f32 product, THEN the rounding to T (not one fused f16 rounding: v_fma_mixlo_f16)
        T q;
        TT<T>::st(&q, pr);
        out[(size_t)n * K + k] = q;
        cs += (double)TT<T>::ld(&q);
        bs += (double)wv * (double)beta[k];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { cs += __shfl_xor(cs, o); bs += __shfl_xor(bs, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = cs; red[1][threadIdx.x >> 6] = bs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double c = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        double bb = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        if (b) bb += (double)b[row];
        colsum[n] = (float)c;
        bias_out[n] = (float)bb;
    }
}

// y[b][iy][ix][:] = x[b][iy / 2][ix / 2][:] on even (iy, ix) inside the source, 0 elsewhere (16-byte chunks): the
// zero-inserted output gradient that turns the data gradient of a stride-2 conv into a stride-1 one
template <typename T>
__global__ __launch_bounds__(256) void zero_insert2x_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int OH,
                                                            int OW, int H, int W, int CPR, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % CPR);
        size_t pix = i / CPR;
        const int ix = (int)(pix % W);
        pix /= W;
        const int iy = (int)(pix % H), b = (int)(pix / H);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (!(iy & 1) && !(ix & 1) && (iy >> 1) < OH && (ix >> 1) < OW)
            v = x[(((size_t)b * OH + (iy >> 1)) * OW + (ix >> 1)) * CPR + q];
        y[i] = v;
    }
}

// y[b][iy][ix][:] = sum of the 2 x 2 block x[b][2 iy + {0, 1}][2 ix + {0, 1}][:] (f32 sum): the gradient of a
// nearest-2x upsample
template <typename T>
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C,
                                                         size_t total) {
    constexpr int EPC = TT<T>::EPC;
    const int CPR = C / EPC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % CPR);
        size_t pix = i / CPR;
        const int ix = (int)(pix % W);
        pix /= W;
        const int iy = (int)(pix % H), b = (int)(pix / H);
        const T* src = x + ((((size_t)b * 2 * H + 2 * iy) * 2 * W) + 2 * ix) * C + q * EPC;
        float a[EPC], f[EPC];
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(src), a);
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(src + C), f);
#pragma unroll
        for (int j = 0; j < EPC; ++j) a[j] += f[j];
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(src + (size_t)2 * W * C), f);
#pragma unroll
        for (int j = 0; j < EPC; ++j) a[j] += f[j];
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(src + (size_t)2 * W * C + C), f);
#pragma unroll
        for (int j = 0; j < EPC; ++j) a[j] += f[j];
        *reinterpret_cast<uint4*>(y + i * EPC) = f32_to_chunk<T>(a);
    }
}

// dx = dy * silu'(x)
template <typename T>
__global__ void silu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float z = TT<T>::ld(x + i);
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        TT<T>::st(dx + i, TT<T>::ld(dy + i) * sg * (1.0f + z * (1.0f - sg)));
    }
}

// y = a + b (gradient accumulation where a tensor feeds two consumers: UNet skip connections, feature taps)
template <typename T>
__global__ void add_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ y, size_t chunks) {
    constexpr int EPC = TT<T>::EPC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (size_t)gridDim.x * blockDim.x) {
        float fa[EPC], fb[EPC];
        chunk_to_f32<T>(a[i], fa);
        chunk_to_f32<T>(b[i], fb);
#pragma unroll
        for (int j = 0; j < EPC; ++j) fa[j] += fb[j];
        y[i] = f32_to_chunk<T>(fa);
    }
}

// out[b][c] += sum over the HW rows of image b of x[b * HW + r][c]: bias / time-row gradients.  grid (column chunks / 256,
// row slices, B); a thread owns one 16-byte column chunk, sums its slice in f32 and adds it with float atomics.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int ldx, int HW, int C, int rows_per_slice,
                                                     float* __restrict__ out) {
    constexpr int EPC = TT<T>::EPC;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q * EPC >= C) return;
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_slice;
    int r1 = r0 + rows_per_slice;
    if (r1 > HW) r1 = HW;
    float s[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) s[j] = 0.f;
    const T* xb = x + ((size_t)b * HW) * ldx + q * EPC;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {   // four independent row loads in flight
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(xb + (size_t)(r + u) * ldx);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float f[EPC];
            chunk_to_f32<T>(v[u], f);
#pragma unroll
            for (int j = 0; j < EPC; ++j) s[j] += f[j];
        }
    }
    for (; r < r1; ++r) {
        float f[EPC];
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(xb + (size_t)r * ldx), f);
#pragma unroll
        for (int j = 0; j < EPC; ++j) s[j] += f[j];
    }
#pragma unroll
    for (int j = 0; j < EPC; ++j) unsafeAtomicAdd(out + (size_t)b * C + q * EPC + j, s[j]);
}

unsigned bwd_grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b ? b : 1));
}

}  // namespace

extern "C" int madm_zero_insert2x(int dtype, const void* x, void* y, int B, int OH, int OW, int H, int W, int C,
                                  void* stream) {
    MADM_REQUIRE(x && y && B > 0 && OH > 0 && OW > 0 && H > 0 && W > 0 && C > 0, "zero_insert2x: bad argument");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(madm_dtype_ok(dtype), "zero_insert2x: unknown dtype %d", dtype);
    MADM_REQUIRE(C % epc == 0, "zero_insert2x: C = %d must be a multiple of %d", C, epc);
    const size_t total = (size_t)B * H * W * (C / epc);
    zero_insert2x_kernel<float><<<bwd_grid_for(total), 256, 0, (hipStream_t)stream>>>((const uint4*)x, (uint4*)y, OH, OW, H,
                                                                                      W, C / epc, total);
    return madm_check_launch("zero_insert2x_kernel");
}

extern "C" int madm_sumpool2x2(int dtype, const void* x, void* y, int B, int H, int W, int C, void* stream) {
    MADM_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0, "sumpool2x2: bad argument");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0, "sumpool2x2: C = %d must be a multiple of %d", C, epc);
    const size_t total = (size_t)B * H * W * (C / epc);
    MADM_DISPATCH_DTYPE(dtype, (sumpool2x2_kernel<T><<<bwd_grid_for(total), 256, 0, (hipStream_t)stream>>>(
                                   (const T*)x, (T*)y, H, W, C, total)));
    return madm_check_launch("sumpool2x2_kernel");
}

extern "C" int madm_colsum(int dtype, const void* x, int ldx, int B, int HW, int C, float* out, void* stream) {
    MADM_REQUIRE(x && out && B > 0 && HW > 0 && C > 0, "colsum: bad argument");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0 && ldx % epc == 0 && ldx >= C, "colsum: C / ldx must be multiples of %d elements", epc);
    int slices = (HW + 31) / 32;
    if (slices > 512) slices = 512;
    const int rows_per_slice = (HW + slices - 1) / slices;
    slices = (HW + rows_per_slice - 1) / rows_per_slice;
    MADM_REQUIRE(B <= 65535, "colsum: too many images");
    dim3 grid((unsigned)((C / epc + 255) / 256), (unsigned)slices, (unsigned)B);
    MADM_DISPATCH_DTYPE(dtype, (colsum_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, ldx, HW, C,
                                                                                        rows_per_slice, out)));
    return madm_check_launch("colsum_kernel");
}

extern "C" int madm_add(int dtype, const void* a, const void* b, void* y, size_t n, void* stream) {
    MADM_REQUIRE(a && b && y && n > 0, "add: bad argument");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(n % epc == 0, "add: n must be a multiple of %d elements", epc);
    const size_t chunks = n / epc;
    MADM_DISPATCH_DTYPE(dtype, (add_kernel<T><<<bwd_grid_for(chunks), 256, 0, (hipStream_t)stream>>>(
                                   (const uint4*)a, (const uint4*)b, (uint4*)y, chunks)));
    return madm_check_launch("add_kernel");
}

extern "C" int madm_silu_bwd(int dtype, const void* x, const void* dy, void* dx, size_t n, void* stream) {
    MADM_REQUIRE(x && dy && dx && n > 0, "silu_bwd: bad argument");
    MADM_DISPATCH_DTYPE(dtype, (silu_bwd_kernel<T><<<bwd_grid_for(n), 256, 0, (hipStream_t)stream>>>(
                                   (const T*)x, (const T*)dy, (T*)dx, n)));
    return madm_check_launch("silu_bwd_kernel");
}

extern "C" int madm_conv2d_wgrad(const madm_conv2d_wgrad_args* a, void* stream) {
    MADM_REQUIRE(a && a->in1 && a->dout && a->dw, "conv2d_wgrad: null argument");
    MADM_REQUIRE(madm_dtype_ok(a->dtype), "conv2d_wgrad: unknown dtype %d", a->dtype);
    const int epc = madm_epc(a->dtype);
    const size_t esz = madm_esize(a->dtype);
    MADM_REQUIRE(a->C1 > 0 && a->C2 >= 0 && (a->C2 == 0 || a->in2), "conv2d_wgrad: bad channel split %d + %d", a->C1,
                 a->C2);
    MADM_REQUIRE(a->C1 % epc == 0 && a->C2 % epc == 0, "conv2d_wgrad: C1 / C2 must be multiples of %d elements", epc);
    MADM_REQUIRE(a->N > 0 && a->N % 4 == 0, "conv2d_wgrad: N = %d must be a positive multiple of 4", a->N);
    MADM_REQUIRE(a->B > 0 && a->IH > 0 && a->IW > 0 && a->OH > 0 && a->OW > 0 && a->KH > 0 && a->KW > 0 &&
                     a->stride > 0,
                 "conv2d_wgrad: bad geometry");
    WgradP p;
    p.in1 = (const char*)a->in1; p.in2 = (const char*)a->in2; p.dout = (const char*)a->dout; p.dw = a->dw;
    p.C1 = a->C1; p.C2 = a->C2; p.Ctot = a->C1 + a->C2;
    p.ld1 = a->ld1 ? a->ld1 : a->C1;
    p.ld2 = a->ld2 ? a->ld2 : a->C2;
    p.ldd = a->ldd ? a->ldd : a->N;
    MADM_REQUIRE(p.ld1 % epc == 0 && p.ld2 % epc == 0 && p.ldd % 4 == 0, "conv2d_wgrad: row strides must keep 16-byte "
                 "(dout: 8-byte) alignment");
    p.B = a->B; p.IH = a->IH; p.IW = a->IW; p.OH = a->OH; p.OW = a->OW; p.KH = a->KH; p.KW = a->KW;
    p.stride = a->stride; p.pad_t = a->pad_t; p.pad_l = a->pad_l; p.upsample = a->upsample ? 1 : 0;
    p.N = a->N; p.K = a->KH * a->KW * p.Ctot;
    p.dbias = a->dbias;
    const long long M = (long long)a->B * a->OH * a->OW;
    const long long in_rows = (long long)a->B * a->IH * a->IW;
    MADM_REQUIRE(M < (1ll << 30) && in_rows * p.ld1 * (long long)esz < (1ll << 31) &&
                     in_rows * (long long)p.ld2 * (long long)esz < (1ll << 31) &&
                     M * p.ldd * (long long)esz < (1ll << 31),
                 "conv2d_wgrad: tensors must stay below 2 GiB");
    p.M = (int)M;
    p.bytes1 = (unsigned)(in_rows * p.ld1 * esz);
    p.bytes2 = a->C2 ? (unsigned)(in_rows * p.ld2 * esz) : 0u;
    p.bytesd = (unsigned)(M * p.ldd * esz);
    p.lin = (a->KH == 1 && a->KW == 1 && a->stride == 1 && a->pad_t == 0 && a->pad_l == 0 && !a->upsample &&
             a->OH == a->IH && a->OW == a->IW)
                ? 1 : 0;
    const int px = 4 * epc;
    const int total_steps = (p.M + px - 1) / px;
    const int tilesN = (p.N + 127) / 128, tilesK = (p.K + 127) / 128;
    int splitm = a->splitm;
    if (splitm <= 0) {
        // one round of resident blocks (2 per CU, a little oversubscribed): every further slice only adds a pass of
        // float atomics over dw (~3 ps per element).  tools/bench_backward.py --sweep: within ~10 % of the best
        // slice count on the conv / linear shapes of the 512 x 512 forward
        const int tiles = tilesN * tilesK;
        splitm = (640 + tiles / 2) / tiles;
        const int cap = total_steps / 2;
        if (splitm > cap) splitm = cap;
        if (splitm < 1) splitm = 1;
        if (total_steps <= 512) {
            // Short reductions (the UNet's 8^2 .. 64^2 maps): a step costs ~1 us whatever the tile (one operand round trip:
            // the loop prefetches one step ahead), a slice's partial tile 16 384 float atomics at ~2.9 ps each, a sole
            // owner's read-modify-write ~4.2 ps per element.  Model  rounds(s) * steps / s * 1 us + atomics(s)  fitted to
            // tools/exp/sweep_wgrad_splitm.py (M 2048 x 640 x 640: 37.5 us at the rule above (25 slices) -> 22.2 at 8;
            // M 512 x 1280 x 1280: 37.7 -> 21.6 at 2; M 512 x 1280 x 5120: 63.8 -> 42.7 at 1)
            double best = 1e30;
            int bs = 1;
            for (int sm = 1; sm <= 32 && sm <= total_steps; ++sm) {
                const int per = (total_steps + sm - 1) / sm;
                const int blocks = tiles * ((total_steps + per - 1) / per);
                const int rounds = (blocks + 511) / 512;
                const double t = (double)rounds * per * 1.0 + (double)tiles * 16384.0 * (sm > 1 ? sm * 2.9e-6 : 4.2e-6);
                if (t < best) { best = t; bs = sm; }
            }
            splitm = bs;
        }
    }
    if (splitm > total_steps) splitm = total_steps;
    MADM_REQUIRE(splitm <= 65535 && tilesK <= 65535, "conv2d_wgrad: grid too large");
    p.steps_per_slice = (total_steps + splitm - 1) / splitm;
    splitm = (total_steps + p.steps_per_slice - 1) / p.steps_per_slice;
    dim3 grid((unsigned)tilesN, (unsigned)tilesK, (unsigned)splitm);
    hipStream_t s = (hipStream_t)stream;
    if (a->dtype == MADM_BF16) conv2d_wgrad_kernel<bf16_t><<<grid, 256, 0, s>>>(p);
    else if (a->dtype == MADM_F16) conv2d_wgrad_kernel<f16_t><<<grid, 256, 0, s>>>(p);
    else conv2d_wgrad_kernel<float><<<grid, 256, 0, s>>>(p);
    return madm_check_launch("conv2d_wgrad_kernel");
}

extern "C" int madm_pack_weight(int dtype, const float* w, void* out, int ldo, int N, int Cin, int taps, int nsrc,
                               const int* src_channels, int ktile, int interleave, void* stream) {
    MADM_REQUIRE(w && out && src_channels && N > 0 && Cin > 0, "pack_weight: bad argument");
    MADM_REQUIRE(madm_dtype_ok(dtype), "pack_weight: unknown dtype %d", dtype);
    MADM_REQUIRE(taps >= 1 && taps <= 9 && nsrc >= 1 && nsrc <= 4 && ktile > 0, "pack_weight: taps <= 9, 1 .. 4 sources");
    MADM_REQUIRE(!interleave || N % 2 == 0, "pack_weight: interleaved halves need an even row count");
    PackP p;
    p.w = w; p.out = out; p.N = N; p.Cin = Cin; p.taps = taps; p.nsrc = nsrc; p.ldo = ldo; p.interleave = interleave ? 1 : 0;
    int c0 = 0, cp0 = 0;
    unsigned blocks = 0;
    for (int s = 0; s < 4; ++s) {
        p.C[s] = s < nsrc ? src_channels[s] : 0;
        MADM_REQUIRE(s >= nsrc || p.C[s] > 0, "pack_weight: empty source");
        p.coff[s] = c0; p.cpoff[s] = cp0;
        p.cpad[s] = (p.C[s] + ktile - 1) / ktile * ktile;
        c0 += p.C[s]; cp0 += p.cpad[s];
        if (s < nsrc) blocks += (unsigned)((p.cpad[s] + 255) / 256);
    }
    MADM_REQUIRE(c0 == Cin, "pack_weight: the sources hold %d channels, the weight %d", c0, Cin);
    p.cpad_tot = cp0;
    MADM_REQUIRE(ldo >= taps * cp0, "pack_weight: ldo = %d < %d packed columns", ldo, taps * cp0);
    MADM_REQUIRE(blocks <= 65535, "pack_weight: grid too large");
    dim3 grid((unsigned)N, blocks);
    MADM_DISPATCH_DTYPE(dtype, (pack_weight_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(p)));
    return madm_check_launch("pack_weight_kernel");
}

extern "C" int madm_fold_layernorm_pack(int dtype, const float* w, const float* b, const float* gamma, const float* beta,
                                       void* out, float* bias_out, float* colsum, int N, int K, int interleave,
                                       void* stream) {
    MADM_REQUIRE(w && gamma && beta && out && bias_out && colsum && N > 0 && K > 0, "fold_layernorm_pack: bad argument");
    MADM_REQUIRE(madm_dtype_ok(dtype), "fold_layernorm_pack: unknown dtype %d", dtype);
    MADM_REQUIRE(!interleave || N % 2 == 0, "fold_layernorm_pack: interleaved halves need an even row count");
    MADM_DISPATCH_DTYPE(dtype, (fold_layernorm_kernel<T><<<(unsigned)N, 256, 0, (hipStream_t)stream>>>(
                                   w, b, gamma, beta, (T*)out, bias_out, colsum, N, K, interleave ? 1 : 0)));
    return madm_check_launch("fold_layernorm_kernel");
}

extern "C" int madm_pack_dgrad_weights(int dtype, const void* w, void* wt, int N, int taps, int C, void* stream) {
    MADM_REQUIRE(w && wt && N > 0 && taps > 0 && C > 0, "pack_dgrad_weights: bad argument");
    MADM_REQUIRE(taps <= 65535 && (N + 31) / 32 <= 65535, "pack_dgrad_weights: grid too large");
    dim3 grid((unsigned)((C + 31) / 32), (unsigned)((N + 31) / 32), (unsigned)taps);
    MADM_DISPATCH_DTYPE(dtype, (pack_dgrad_weights_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(
                                   (const T*)w, (T*)wt, N, taps, C)));
    return madm_check_launch("pack_dgrad_weights_kernel");
}
