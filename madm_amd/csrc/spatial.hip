// Spatial glue of the projection / segmentation-head rows of the path:
// bilinear resize (tokens and NCHW images), depthwise dilated 3x3 conv (+ folded BatchNorm + ReLU),
// tanh-gated prompt / time conditioning, per-pixel argmax.  All HBM-bound streaming kernels.
#include "common.hpp"

namespace {

// PyTorch F.interpolate(mode='bilinear', align_corners=False) source index / weight
__device__ __forceinline__ void bilinear_coord(int o, int in_size, float scale, int& i0, int& i1, float& l1) {
    float src = ((float)o + 0.5f) * scale - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

// channels-last: in [B*IH*IW][ldi] (C channels) -> out [B*OH*OW][ldo] (out already points at its column offset)
template <typename T>
__global__ __launch_bounds__(256) void resize_bilinear_tokens_kernel(const T* __restrict__ in, int ldi, T* __restrict__ out,
                                                                     int ldo, int B, int IH, int IW, int OH, int OW,
                                                                     int C) {
    constexpr int EPC = TT<T>::EPC;
    const unsigned CPR = (unsigned)C / EPC;
    const unsigned total = (unsigned)B * OH * OW * CPR;
    const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned q = idx % CPR;
        const unsigned pix = idx / CPR;
        const unsigned ox = pix % (unsigned)OW;
        const unsigned t = pix / (unsigned)OW;
        const unsigned oy = t % (unsigned)OH;
        const unsigned b = t / (unsigned)OH;
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_coord((int)oy, IH, sy, y0, y1, ly);
        bilinear_coord((int)ox, IW, sx, x0, x1, lx);
        const T* base = in + (size_t)b * IH * IW * ldi + q * EPC;
        float a[EPC], c[EPC], d[EPC], e[EPC], o[EPC];
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(base + ((size_t)y0 * IW + x0) * ldi), a);
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(base + ((size_t)y0 * IW + x1) * ldi), c);
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(base + ((size_t)y1 * IW + x0) * ldi), d);
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(base + ((size_t)y1 * IW + x1) * ldi), e);
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
        for (int j = 0; j < EPC; ++j) o[j] = w00 * a[j] + w01 * c[j] + w10 * d[j] + w11 * e[j];
        *reinterpret_cast<uint4*>(out + (size_t)pix * ldo + q * EPC) = f32_to_chunk<T>(o);
    }
}

// NCHW f32 planes
__global__ __launch_bounds__(256) void resize_bilinear_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                   int planes, int IH, int IW, int OH, int OW) {
    const size_t total = (size_t)planes * OH * OW;
    const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % OW);
        const size_t t = idx / OW;
        const int oy = (int)(t % OH);
        const size_t pl = t / OH;
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_coord(oy, IH, sy, y0, y1, ly);
        bilinear_coord(ox, IW, sx, x0, x1, lx);
        const float* p = in + pl * IH * IW;
        const float top = p[(size_t)y0 * IW + x0] * (1.f - lx) + p[(size_t)y0 * IW + x1] * lx;
        const float bot = p[(size_t)y1 * IW + x0] * (1.f - lx) + p[(size_t)y1 * IW + x1] * lx;
        out[idx] = top * (1.f - ly) + bot * ly;
    }
}

// depthwise 3x3, dilation d, padding d, stride 1; w [9][C] f32 (tap-major), then y = act(acc * scale[c] + shift[c])
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        T* __restrict__ y, int ldy, int B, int H, int W, int C, int dil,
                                                        int act) {
    constexpr int EPC = TT<T>::EPC;
    const unsigned CPR = (unsigned)C / EPC;
    const unsigned total = (unsigned)B * H * W * CPR;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned q = idx % CPR;
        const unsigned pix = idx / CPR;
        const int ox = (int)(pix % (unsigned)W);
        const unsigned t = pix / (unsigned)W;
        const int oy = (int)(t % (unsigned)H);
        const unsigned b = t / (unsigned)H;
        float acc[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) acc[j] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oy + (r - 1) * dil;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ix = ox + (s - 1) * dil;
                if ((unsigned)ix >= (unsigned)W) continue;
                float f[EPC];
                chunk_to_f32<T>(*reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * C + q * EPC), f);
                const float* wt = w + (size_t)(r * 3 + s) * C + q * EPC;
#pragma unroll
                for (int j = 0; j < EPC; ++j) acc[j] += f[j] * wt[j];
            }
        }
#pragma unroll
        for (int j = 0; j < EPC; ++j) acc[j] = acc[j] * scale[q * EPC + j] + shift[q * EPC + j];
        act_inplace<EPC>(acc, act);
        *reinterpret_cast<uint4*>(y + (size_t)pix * ldy + q * EPC) = f32_to_chunk<T>(acc);
    }
}

// The same conv with PX outputs per thread, spaced one DILATION apart along x (x = r + dil * k): neighbouring outputs of
// that comb share their taps, so the thread loads (PX + 2) x 3 input chunks for PX outputs -- 4.5 loads per output at
// PX = 4 instead of 9.  The plain kernel re-fetches every input element nine times through the L1 / L2 path and ran at
// 1.85 ms for the head's 1 GB tensor (0.5 ms of HBM time); used when the map is at least 4 dilations wide.
//
// XCD: workgroups go round-robin to the 8 XCDs, so the threads that share an input element -- the neighbours one dilation
// up / down / sideways -- sat behind eight different L2s and every one of the 4.5 requests per element crossed the fabric.
// With XCD = true workgroup g serves channel slab g % 8 (C / 8 channels, >= 128 B per pixel) of ALL pixels: whatever
// re-reads an element runs on the same XCD, a few rows away in time.  Head tensor (512 x 512 x 1024 f16, 0.5 GB): 450 ->
// 390 us at dilation 6, 430 -> 400 at 12, 460 -> 430 at 18 (tools/exp/dwconv_ab.py).  Rejected on the same tensor: a
// column walk with the three input rows held in registers (1.5 requests per element, but 256 VGPRs: 670 us) and
// requesting all 18 chunks up front with the taps in LDS (214 VGPRs, two workgroups per CU: 450 us).
template <typename T, int PX, bool XCD>
__global__ __launch_bounds__(256) void dwconv3x3_comb_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             T* __restrict__ y, int ldy, int B, int H, int W, int C, int dil,
                                                             int act, int kblocks) {
    constexpr int EPC = TT<T>::EPC;
    const unsigned CPR = (unsigned)C / EPC;
    const unsigned QL = XCD ? CPR / 8 : CPR;                      // channel chunks per slab
    const unsigned slab = XCD ? (blockIdx.x & 7u) : 0u;
    const size_t total = (size_t)B * H * dil * kblocks * QL;      // threads' worth of work per slab
    const size_t first = (size_t)(XCD ? blockIdx.x >> 3 : blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)(XCD ? gridDim.x >> 3 : gridDim.x) * blockDim.x;
    for (size_t idx = first; idx < total; idx += stride) {
        const unsigned q = slab * QL + (unsigned)(idx % QL);
        size_t t = idx / QL;
        const int kb = (int)(t % (unsigned)kblocks); t /= (unsigned)kblocks;
        const int r = (int)(t % (unsigned)dil); t /= (unsigned)dil;
        const int oy = (int)(t % (unsigned)H);
        const size_t b = t / (unsigned)H;
        const int x0 = r + dil * (kb * PX);           // first output column of this thread
        if (x0 >= W) continue;
        float acc[PX][EPC];
#pragma unroll
        for (int o = 0; o < PX; ++o)
#pragma unroll
            for (int j = 0; j < EPC; ++j) acc[o][j] = 0.f;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int iy = oy + (rr - 1) * dil;
            if ((unsigned)iy >= (unsigned)H) continue;
            float wt[3][EPC];
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int j = 0; j < EPC; ++j) wt[s][j] = w[(size_t)(rr * 3 + s) * C + q * EPC + j];
            const T* row = x + ((b * H + iy) * (size_t)W) * C + q * EPC;
#pragma unroll
            for (int c = 0; c < PX + 2; ++c) {         // input column x0 + (c - 1) * dil feeds outputs c - 2, c - 1, c
                const int ix = x0 + (c - 1) * dil;
                if ((unsigned)ix >= (unsigned)W) continue;
                float f[EPC];
                chunk_to_f32<T>(*reinterpret_cast<const uint4*>(row + (size_t)ix * C), f);
#pragma unroll
                for (int s = 0; s < 3; ++s) {          // tap s reads column o + (s - 1): o = c - s
                    const int o = c - s;
                    if (o >= 0 && o < PX) {
#pragma unroll
                        for (int j = 0; j < EPC; ++j) acc[o][j] += f[j] * wt[s][j];
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < PX; ++o) {
            const int ox = x0 + o * dil;
            if (ox >= W) break;
#pragma unroll
            for (int j = 0; j < EPC; ++j) acc[o][j] = acc[o][j] * scale[q * EPC + j] + shift[q * EPC + j];
            act_inplace<EPC>(acc[o], act);
            *reinterpret_cast<uint4*>(y + (((b * H + oy) * (size_t)W) + ox) * ldy + q * EPC) = f32_to_chunk<T>(acc[o]);
        }
    }
}

// The dilated conv seen on its own lattice: the pixels (ry + d i, rx + d j) of one residue class form a dense
// ceil(H / d) x ceil(W / d) image on which the conv is a plain 3x3.  A workgroup takes a tile of at most 15 x 15 outputs
// of one residue class and one 128-byte channel slab (8 x 16-byte chunks), stages the (th + 2) x (tw + 2) halo in LDS
// -- every pixel of it one full 128-byte line -- and each thread forms 8 horizontally adjacent outputs of one chunk
// from 3 x 10 LDS reads: 1.27 global requests per input element instead of the comb kernel's 4.5 (which all went
// through the L1 / L2 path: 390 .. 430 us for the head's 0.5 GB tensor against 200 us of HBM time).  Taps in the order
// of the other two kernels (row-major, fused multiply-add): identical bits.  grid.x = slab fastest, then tile, residue,
// image: the 16 slabs of a pixel (2 KB at C = 1024) are requested by neighbouring workgroups at about the same time.
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_lattice_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                T* __restrict__ y, int ldy, int B, int H, int W, int C, int dil,
                                                                int act, int ny, int nx, int th, int tw) {
    constexpr int EPC = TT<T>::EPC;
    constexpr int CHS = 8 * EPC;                       // channels per slab (128 bytes)
    __shared__ __attribute__((aligned(16))) uint4 halo[17 * 17 * 8];   // tiles of at most 15 x 15: 37 KB, four workgroups per CU
    __shared__ __attribute__((aligned(16))) float wl[11 * CHS];   // 9 taps, scale, shift of the slab's channels
    const int tid = threadIdx.x;
    unsigned t = blockIdx.x;
    const int slabs = C / CHS;
    const int slab = (int)(t % (unsigned)slabs); t /= (unsigned)slabs;
    const int tx = (int)(t % (unsigned)nx); t /= (unsigned)nx;
    const int ty = (int)(t % (unsigned)ny); t /= (unsigned)ny;
    const int rx = (int)(t % (unsigned)dil); t /= (unsigned)dil;
    const int ry = (int)(t % (unsigned)dil);
    const int b = (int)(t / (unsigned)dil);
    const int c0 = slab * CHS;
    const int sy0 = ty * th, sx0 = tx * tw;            // first lattice row / column of the tile
    const int hw_ = tw + 2, items = (th + 2) * hw_ * 8;
    // ---- halo: all of a thread's pieces requested before the first is written ----
    constexpr int MAXP = (17 * 17 * 8 + 255) / 256;    // 10
    uint4 hv[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int i = tid + 256 * k;
        hv[k] = make_uint4(0u, 0u, 0u, 0u);
        if (i < items) {
            const int hp = i >> 3, q = i & 7;
            const int hy = hp / hw_, hx = hp - hy * hw_;
            const int iy = ry + dil * (sy0 + hy - 1), ix = rx + dil * (sx0 + hx - 1);
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                hv[k] = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * C + c0 + q * EPC);
        }
    }
    float4 wv4 = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int row = tid / (CHS / 4), col = (tid % (CHS / 4)) * 4;
        const float* src = row < 9 ? w + (size_t)row * C : (row == 9 ? scale : shift);
        if (row < 11) wv4 = *reinterpret_cast<const float4*>(src + c0 + col);
    }
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int i = tid + 256 * k;
        if (i < items) halo[i] = hv[k];
    }
    if (tid < 11 * CHS / 4) *reinterpret_cast<float4*>(wl + tid * 4) = wv4;
    __syncthreads();
    // ---- 8 outputs of one chunk per thread: row r of the tile, columns half * 8 .. + 7 ----
    const int q = tid & 7, g = tid >> 3;
    const int r = g >> 1, half = g & 1;
    const int oy = ry + dil * (sy0 + r);
    if (r >= th || half * 8 >= tw || oy >= H) return;
    float acc[8][EPC];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int j = 0; j < EPC; ++j) acc[o][j] = 0.f;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        const int iy = oy + (rr - 1) * dil;
        if ((unsigned)iy < (unsigned)H) {              // (taps outside the image are skipped, as in the other kernels)
        float wt[3][EPC];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
            for (int j = 0; j < EPC; ++j) wt[s_][j] = wl[(rr * 3 + s_) * CHS + q * EPC + j];
        const uint4* hrow = halo + ((r + rr) * hw_ + half * 8) * 8 + q;
#pragma unroll
        for (int c = 0; c < 10; ++c) {                 // halo column half * 8 + c feeds outputs c - 2, c - 1, c
            // (no break / continue in here: the loop must unroll completely, acc[] is indexed by c - s)
            const int ix = rx + dil * (sx0 + half * 8 + c - 1);
            if (half * 8 + c < hw_ && (unsigned)ix < (unsigned)W) {
                float f[EPC];
                chunk_to_f32<T>(hrow[c * 8], f);
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) {
                    if (c - s_ >= 0 && c - s_ < 8) {
#pragma unroll
                        for (int j = 0; j < EPC; ++j) acc[c - s_][j] += f[j] * wt[s_][j];
                    }
                }
            }
        }
        }
    }
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) { sc[j] = wl[9 * CHS + q * EPC + j]; sh[j] = wl[10 * CHS + q * EPC + j]; }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        const int sx = half * 8 + o;
        const int ox = rx + dil * (sx0 + sx);
        if (sx < tw && ox < W) {
#pragma unroll
            for (int j = 0; j < EPC; ++j) acc[o][j] = acc[o][j] * sc[j] + sh[j];
            act_inplace<EPC>(acc[o], act);
            *reinterpret_cast<uint4*>(y + (((size_t)b * H + oy) * W + ox) * ldy + c0 + q * EPC) = f32_to_chunk<T>(acc[o]);
        }
    }
}

// out[pl][y][x] = in[pl][y1 + y][x1 + x] * scale inside the input, 0 outside: zero padding, cropping, windows
__global__ void scale_pad_crop_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int IH, int IW,
                                      int y1, int x1, int OH, int OW, float scale) {
    const size_t total = (size_t)planes * OH * OW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % OW) + x1;
        const size_t t = idx / OW;
        const int y = (int)(t % OH) + y1;
        const size_t pl = t / OH;
        out[idx] = ((unsigned)y < (unsigned)IH && (unsigned)x < (unsigned)IW) ? in[(pl * IH + y) * IW + x] * scale : 0.f;
    }
}

// sliding-window merge (feature_extractor.py:199-278): the windows' feature maps [nW][B][h][w][C] (channels-last,
// window-major) are averaged into the canvas [B][h][Wc][C]: out = sum_w win_w / count, window w covering columns
// [x1[w], x1[w] + w) of the canvas
template <typename T>
__global__ __launch_bounds__(256) void slide_merge_kernel(const T* __restrict__ win, T* __restrict__ out, int nW, int B,
                                                          int h, int w, int Wc, int C, int4 x1) {
    constexpr int EPC = TT<T>::EPC;
    const unsigned CPR = (unsigned)C / EPC;
    const unsigned total = (unsigned)B * h * Wc * CPR;
    const int xs[4] = {x1.x, x1.y, x1.z, x1.w};
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned q = idx % CPR;
        const unsigned pix = idx / CPR;
        const int X = (int)(pix % (unsigned)Wc);
        const unsigned t = pix / (unsigned)Wc;
        const int y = (int)(t % (unsigned)h);
        const int b = (int)(t / (unsigned)h);
        float acc[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) acc[j] = 0.f;
        int cnt = 0;
        for (int k = 0; k < nW; ++k) {
            const int lx = X - xs[k];
            if ((unsigned)lx < (unsigned)w) {
                float f[EPC];
                chunk_to_f32<T>(*reinterpret_cast<const uint4*>(win + ((((size_t)k * B + b) * h + y) * w + lx) * C + q * EPC), f);
#pragma unroll
                for (int j = 0; j < EPC; ++j) acc[j] += f[j];
                ++cnt;
            }
        }
        const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
#pragma unroll
        for (int j = 0; j < EPC; ++j) acc[j] *= inv;
        *reinterpret_cast<uint4*>(out + (size_t)pix * C + q * EPC) = f32_to_chunk<T>(acc);
    }
}

// confusion matrix of the evaluator (evaluation/d2_evaluator.py:106-127): conf[(K+1) * pred + gt'] += 1 with
// gt' = K where gt == ignore_label.  Per-block LDS histogram, 64-bit global counters: exact integer arithmetic.
__global__ __launch_bounds__(256) void confusion_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ gt,
                                                        size_t n, int K, int ignore_label,
                                                        unsigned long long* __restrict__ conf) {
    extern __shared__ unsigned int hist[];
    const int bins = (K + 1) * (K + 1);
    for (int i = threadIdx.x; i < bins; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        long long g = gt[i];
        if (g == ignore_label) g = K;
        const long long p = pred[i];
        if (p >= 0 && p <= K && g >= 0 && g <= K) atomicAdd(&hist[(K + 1) * (int)p + (int)g], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += blockDim.x)
        if (hist[i]) atomicAdd(&conf[i], (unsigned long long)hist[i]);
}

__global__ void tanh_gate_kernel(const float* __restrict__ a1, const float* __restrict__ x1, const float* __restrict__ a2,
                                 const float* __restrict__ x2, float* __restrict__ out, size_t n, int repeat) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = (a1 ? tanhf(a1[i]) : 1.f) * x1[i];
        if (x2) v += (a2 ? tanhf(a2[i]) : 1.f) * x2[i];
        for (int r = 0; r < repeat; ++r) out[(size_t)r * n + i] = v;
    }
}

// NCHW f32 logits [B][K][HW] -> int64 labels [B][HW]; first maximal channel wins (torch.argmax semantics)
__global__ void argmax_nchw_kernel(const float* __restrict__ x, int64_t* __restrict__ out, int B, int K, size_t HW) {
    const size_t total = (size_t)B * HW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t b = idx / HW, p = idx - b * HW;
        const float* px = x + b * K * HW + p;
        float best = px[0];
        int bi = 0;
        for (int k = 1; k < K; ++k) {
            const float v = px[(size_t)k * HW];
            if (v > best || (v != v && best == best)) { best = v; bi = k; }   // NaN counts as maximal, like torch
        }
        out[idx] = bi;
    }
}

unsigned grid_for(size_t n, unsigned cap = 4096) {
    size_t g = (n + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" {

int madm_resize_bilinear(int dtype, const void* in, int ldi, void* out, int ldo, int B, int IH, int IW, int OH, int OW,
                         int C, void* stream) {
    MADM_REQUIRE(in && out && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0, "resize_bilinear: bad args");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0 && ldi >= C && ldo >= C && ldi % epc == 0 && ldo % epc == 0, "resize_bilinear: C/ld must be multiples of %d", epc);
    const size_t total = (size_t)B * OH * OW * (C / epc);
    MADM_REQUIRE(total < 0x7fffffffull, "resize_bilinear: tensor too large for 32-bit indexing");
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (resize_bilinear_tokens_kernel<T><<<grid_for(total), 256, 0, s>>>(
                                   (const T*)in, ldi, (T*)out, ldo, B, IH, IW, OH, OW, C)));
    return madm_check_launch("resize_bilinear_tokens_kernel");
}

int madm_resize_bilinear_nchw_f32(const float* in, float* out, int planes, int IH, int IW, int OH, int OW, void* stream) {
    MADM_REQUIRE(in && out && planes > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "resize_bilinear_nchw: bad args");
    resize_bilinear_nchw_kernel<<<grid_for((size_t)planes * OH * OW), 256, 0, (hipStream_t)stream>>>(in, out, planes, IH,
                                                                                                   IW, OH, OW);
    return madm_check_launch("resize_bilinear_nchw_kernel");
}

int madm_dwconv3x3(int dtype, const void* x, const float* w, const float* scale, const float* shift, void* y, int ldy,
                   int B, int H, int W, int C, int dilation, int act, void* stream) {
    MADM_REQUIRE(x && w && scale && shift && y && B > 0 && H > 0 && W > 0 && C > 0 && dilation > 0, "dwconv3x3: bad args");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0 && ldy >= C && ldy % epc == 0 && act >= 0 && act <= 2, "dwconv3x3: bad C/ldy/act");
    const size_t total = (size_t)B * H * W * (C / epc);
    MADM_REQUIRE(total < 0x7fffffffull, "dwconv3x3: tensor too large for 32-bit indexing");
    hipStream_t s = (hipStream_t)stream;
    // MADM_DWCONV_KERNEL (tests, A/B runs): 0 / unset = choose, 1 = plain, 2 = comb, 3 = comb with per-XCD channel slabs, 4 = lattice
    const char* fe = getenv("MADM_DWCONV_KERNEL");
    const int force = fe ? atoi(fe) : 0;
    // 4 = the lattice kernel (one residue class of the dilation per workgroup, halo in LDS): large maps with whole
    // 128-byte channel slabs
    {
        const int chs = 8 * epc;
        const int shmax = (H + dilation - 1) / dilation, swmax = (W + dilation - 1) / dilation;
        const bool fits = C % chs == 0 && dilation > 1;
        const bool pays = total >= ((size_t)1 << 21) && shmax >= 8 && swmax >= 8;
        if (fits && (force == 4 || (force == 0 && pays))) {
            const int ny = (shmax + 14) / 15, nx = (swmax + 14) / 15;
            const int th = (shmax + ny - 1) / ny, tw = (swmax + nx - 1) / nx;
            const size_t blocks = (size_t)B * dilation * dilation * ny * nx * (C / chs);
            MADM_REQUIRE(blocks < 0x7fffffffull, "dwconv3x3: grid too large");
            MADM_DISPATCH_DTYPE(dtype, (dwconv3x3_lattice_kernel<T><<<(unsigned)blocks, 256, 0, s>>>(
                                           (const T*)x, w, scale, shift, (T*)y, ldy, B, H, W, C, dilation, act, ny, nx, th, tw)));
            return madm_check_launch("dwconv3x3_lattice_kernel");
        }
    }
    if (force != 1 && W >= 4 * dilation) {
        constexpr int PX = 4;
        const int kmax = (W + dilation - 1) / dilation;                  // outputs per residue class (upper bound)
        const int kblocks = (kmax + PX - 1) / PX;
        const size_t tot = (size_t)B * H * dilation * kblocks * (C / epc);
        // slabs of at least 128 B per pixel, and enough work for eight workgroups per CU
        const bool slabs = (C / epc) % 8 == 0 &&
                           (force == 3 || (force == 0 && C / 8 * madm_esize(dtype) >= 128 && tot >= ((size_t)1 << 19)));
        if (slabs) {
            const unsigned g = (grid_for(tot, 65536) + 7u) & ~7u;
            MADM_DISPATCH_DTYPE(dtype, (dwconv3x3_comb_kernel<T, PX, true><<<g, 256, 0, s>>>(
                                           (const T*)x, w, scale, shift, (T*)y, ldy, B, H, W, C, dilation, act, kblocks)));
        } else {
            MADM_DISPATCH_DTYPE(dtype, (dwconv3x3_comb_kernel<T, PX, false><<<grid_for(tot, 65536), 256, 0, s>>>(
                                           (const T*)x, w, scale, shift, (T*)y, ldy, B, H, W, C, dilation, act, kblocks)));
        }
        return madm_check_launch("dwconv3x3_comb_kernel");
    }
    MADM_DISPATCH_DTYPE(dtype, (dwconv3x3_kernel<T><<<grid_for(total, 16384), 256, 0, s>>>((const T*)x, w, scale, shift, (T*)y,
                                                                                        ldy, B, H, W, C, dilation, act)));
    return madm_check_launch("dwconv3x3_kernel");
}

int madm_scale_pad_crop_nchw_f32(const float* in, float* out, int planes, int IH, int IW, int y1, int x1, int OH, int OW,
                                 float scale, void* stream) {
    MADM_REQUIRE(in && out && planes > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "scale_pad_crop: bad args");
    scale_pad_crop_kernel<<<grid_for((size_t)planes * OH * OW), 256, 0, (hipStream_t)stream>>>(in, out, planes, IH, IW, y1,
                                                                                             x1, OH, OW, scale);
    return madm_check_launch("scale_pad_crop_kernel");
}

int madm_slide_merge(int dtype, const void* win, void* out, int nW, int B, int h, int w, int Wc, int C, const int* x1,
                     void* stream) {
    MADM_REQUIRE(win && out && x1 && nW >= 1 && nW <= 4 && B > 0 && h > 0 && w > 0 && Wc >= w && C > 0, "slide_merge: bad args");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0, "slide_merge: C must be a multiple of %d", epc);
    const size_t total = (size_t)B * h * Wc * (C / epc);
    MADM_REQUIRE(total < 0x7fffffffull, "slide_merge: tensor too large for 32-bit indexing");
    int4 xs = make_int4(x1[0], nW > 1 ? x1[1] : 0, nW > 2 ? x1[2] : 0, nW > 3 ? x1[3] : 0);
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (slide_merge_kernel<T><<<grid_for(total), 256, 0, s>>>((const T*)win, (T*)out, nW, B, h, w, Wc,
                                                                                   C, xs)));
    return madm_check_launch("slide_merge_kernel");
}

int madm_confusion_matrix(const int64_t* pred, const int64_t* gt, size_t n, int num_classes, int ignore_label,
                          int64_t* conf, void* stream) {
    MADM_REQUIRE(pred && gt && conf && n > 0 && num_classes > 0 && num_classes < 100, "confusion_matrix: bad args");
    const size_t shm = (size_t)(num_classes + 1) * (num_classes + 1) * sizeof(unsigned int);
    confusion_kernel<<<grid_for(n, 1024), 256, shm, (hipStream_t)stream>>>(pred, gt, n, num_classes, ignore_label,
                                                                           (unsigned long long*)conf);
    return madm_check_launch("confusion_kernel");
}

int madm_tanh_gate(const float* a1, const float* x1, const float* a2, const float* x2, float* out, size_t n, int repeat,
                   void* stream) {
    MADM_REQUIRE(x1 && out && n > 0 && repeat > 0 && (!a2 || x2), "tanh_gate: bad args");
    tanh_gate_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(a1, x1, a2, x2, out, n, repeat);
    return madm_check_launch("tanh_gate_kernel");
}

int madm_argmax_nchw_f32(const float* x, int64_t* out, int B, int K, size_t HW, void* stream) {
    MADM_REQUIRE(x && out && B > 0 && K > 0 && HW > 0, "argmax_nchw: bad args");
    argmax_nchw_kernel<<<grid_for((size_t)B * HW), 256, 0, (hipStream_t)stream>>>(x, out, B, K, HW);
    return madm_check_launch("argmax_nchw_kernel");
}

}  // extern "C"
