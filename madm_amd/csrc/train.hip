// Kernels the training step of the path adds around the conv / norm / attention gradient kernels (SURVEY.md 8 a9, a10,
// 8f rank 2; BASELINE config 4): the train-mode DAFormer head (Dropout2d, depthwise-conv weight gradient, bilinear
// resize adjoint), the CmdiseCriterion losses with their gradients (pixel-weighted cross entropy with ignore index,
// masked L1 / L2 on the latents) and the backward of the tanh-gated prompt / time conditioning.  In the reference all of
// these are torch autograd nodes behind ``losses.backward()`` (engine/train_loop.py:203-217).  HBM-bound streaming kernels.
#include "common.hpp"

namespace {

__device__ __forceinline__ void bilinear_coord(int o, int in_size, float scale, int& i0, int& i1, float& l1) {
    float src = ((float)o + 0.5f) * scale - 0.5f;   // F.interpolate(bilinear, align_corners=False), as spatial.hip
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

// y[b][p][c] = x[b][p][c] * s[b][c]: Dropout2d (s = keep-mask / (1 - p) per (image, channel)); its own backward on dy
template <typename T>
__global__ __launch_bounds__(256) void scale_channels_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ s,
                                                             T* __restrict__ y, int ldy, int HW, int C, size_t total) {
    constexpr int EPC = TT<T>::EPC;
    const unsigned CPR = (unsigned)C / EPC;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const unsigned q = (unsigned)(idx % CPR);
        const size_t pix = idx / CPR;
        const size_t b = pix / (size_t)HW;
        float f[EPC];
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(x + pix * ldx + q * EPC), f);
        const float* sb = s + b * C + q * EPC;
#pragma unroll
        for (int j = 0; j < EPC; ++j) f[j] *= sb[j];
        *reinterpret_cast<uint4*>(y + pix * ldy + q * EPC) = f32_to_chunk<T>(f);
    }
}

// dx = dy where y > 0 (y = the ReLU's OUTPUT), else 0
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx,
                                                       size_t chunks) {
    constexpr int EPC = TT<T>::EPC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (size_t)gridDim.x * blockDim.x) {
        float a[EPC], d[EPC];
        chunk_to_f32<T>(reinterpret_cast<const uint4*>(y)[i], a);
        chunk_to_f32<T>(reinterpret_cast<const uint4*>(dy)[i], d);
#pragma unroll
        for (int j = 0; j < EPC; ++j) d[j] = a[j] > 0.f ? d[j] : 0.f;
        reinterpret_cast<uint4*>(dx)[i] = f32_to_chunk<T>(d);
    }
}

// weight gradient of the depthwise dilated 3x3 conv: dw[t][c] += sum_pixels dy[p][c] * x[p + off_t][c].
// grid = pixel slices; threads = (channel chunk, pixel lane); 9 x EPC register sums per thread -> LDS -> f32 atomics.
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, int lddy,
                                                              float* __restrict__ dw, int B, int H, int W, int C, int dil,
                                                              int pix_per_block) {
    constexpr int EPC = TT<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [9][C]
    for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const int CPR = C / EPC;
    const int cols = CPR < 256 ? CPR : 256;
    const int lanes = 256 / cols;
    const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
    const size_t npix = (size_t)B * H * W;
    const size_t p0 = (size_t)blockIdx.x * pix_per_block;
    size_t p1 = p0 + pix_per_block;
    if (p1 > npix) p1 = npix;
    if (ty < lanes) {
        for (int q = tx; q < CPR; q += cols) {
            float acc[9][EPC];
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < EPC; ++j) acc[t][j] = 0.f;
            for (size_t p = p0 + ty; p < p1; p += lanes) {
                const int ox = (int)(p % (size_t)W);
                const size_t t_ = p / (size_t)W;
                const int oy = (int)(t_ % (size_t)H);
                const size_t b = t_ / (size_t)H;
                float d[EPC];
                chunk_to_f32<T>(*reinterpret_cast<const uint4*>(dy + p * lddy + q * EPC), d);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int iy = oy + (r - 1) * dil;
                    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const int ix = ox + (s - 1) * dil;
                        if ((unsigned)ix >= (unsigned)W) continue;
                        float f[EPC];
                        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(x + ((b * H + iy) * W + ix) * C + q * EPC), f);
#pragma unroll
                        for (int j = 0; j < EPC; ++j) acc[r * 3 + s][j] += d[j] * f[j];
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < EPC; ++j) atomicAdd(&lds[t * C + q * EPC + j], acc[t][j]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) unsafeAtomicAdd(dw + i, lds[i]);
}

// The same gradient on the dilation's lattice (spatial.hip, dwconv3x3_lattice_kernel): a workgroup owns one residue
// class (ry, rx) of one image and one 128-byte channel slab and walks that class's tiles of at most 15 x 15 lattice
// pixels; per tile the x halo sits in LDS (1.27 global requests per element instead of 9), every thread multiplies its 8
// dy chunks with the 3 x 10 halo chunks around them into 9 x EPC register sums that live across the tiles; one LDS +
// global atomic round per workgroup (measurements at the launcher).
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_lattice_kernel(const T* __restrict__ x, const T* __restrict__ dy, int lddy,
                                                                      float* __restrict__ dw, int B, int H, int W, int C, int dil,
                                                                      int ny, int nx, int th, int tw) {
    constexpr int EPC = TT<T>::EPC;
    constexpr int CHS = 8 * EPC;
    __shared__ __attribute__((aligned(16))) uint4 halo[17 * 17 * 8];   // tiles of at most 15 x 15: 37 KB, four workgroups per CU
    __shared__ float red[9 * CHS];
    const int tid = threadIdx.x;
    unsigned t = blockIdx.x;
    const int slabs = C / CHS;
    const int slab = (int)(t % (unsigned)slabs); t /= (unsigned)slabs;
    const int rx = (int)(t % (unsigned)dil); t /= (unsigned)dil;
    const int ry = (int)(t % (unsigned)dil);
    const int b = (int)(t / (unsigned)dil);
    const int c0 = slab * CHS;
    const int hw_ = tw + 2, items = (th + 2) * hw_ * 8;
    const int q = tid & 7, g = tid >> 3;
    const int r = g >> 1, half = g & 1;
    for (int i = tid; i < 9 * CHS; i += 256) red[i] = 0.f;
    float acc[9][EPC];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < EPC; ++j) acc[k][j] = 0.f;
    constexpr int MAXP = (17 * 17 * 8 + 255) / 256;
    for (int tile = 0; tile < ny * nx; ++tile) {
        const int ty = tile / nx, tx = tile - ty * nx;
        const int sy0 = ty * th, sx0 = tx * tw;
        uint4 hv[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int i = tid + 256 * k;
            hv[k] = make_uint4(0u, 0u, 0u, 0u);
            if (i < items) {
                const int hp = i >> 3, qq = i & 7;
                const int hy = hp / hw_, hx = hp - hy * hw_;
                const int iy = ry + dil * (sy0 + hy - 1), ix = rx + dil * (sx0 + hx - 1);
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    hv[k] = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * C + c0 + qq * EPC);
            }
        }
        // the thread's 8 dy chunks: row r of the tile, columns half * 8 .. + 7 (zeros outside the tile / the image)
        const int oy = ry + dil * (sy0 + r);
        const bool row_on = r < th && oy < H;
        uint4 dv[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int sx = half * 8 + o;
            const int ox = rx + dil * (sx0 + sx);
            dv[o] = make_uint4(0u, 0u, 0u, 0u);
            if (row_on && sx < tw && ox < W)
                dv[o] = *reinterpret_cast<const uint4*>(dy + (((size_t)b * H + oy) * W + ox) * lddy + c0 + q * EPC);
        }
        __syncthreads();   // the previous tile's halo has been read by every wave
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int i = tid + 256 * k;
            if (i < items) halo[i] = hv[k];
        }
        __syncthreads();
        if (row_on) {
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const uint4* hrow = halo + ((r + rr) * hw_ + half * 8) * 8 + q;
#pragma unroll
                for (int c = 0; c < 10; ++c) {      // halo column half * 8 + c meets outputs c - s under tap (rr, s)
                    if (half * 8 + c < hw_) {       // (pixels outside the image hold zeros in the halo)
                        float f[EPC];
                        chunk_to_f32<T>(hrow[c * 8], f);
#pragma unroll
                        for (int s_ = 0; s_ < 3; ++s_) {
                            if (c - s_ >= 0 && c - s_ < 8) {
                                float d[EPC];
                                chunk_to_f32<T>(dv[c - s_], d);
#pragma unroll
                                for (int j = 0; j < EPC; ++j) acc[rr * 3 + s_][j] += d[j] * f[j];
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < EPC; ++j) atomicAdd(&red[k * CHS + q * EPC + j], acc[k][j]);
    __syncthreads();
    for (int i = tid; i < 9 * CHS; i += 256) unsafeAtomicAdd(dw + (size_t)(i / CHS) * C + c0 + (i % CHS), red[i]);
}

// adjoint of the 1-D bilinear resize along one axis of a [outer][L][inner] tensor:
//   out[o][i][r] = sum_l w(l -> i) * in[o][l][r],   l over the Lout positions whose interpolation reads source i.
// IN_T / OUT_T: the side stored as T (16-byte chunks of EPC elements), the other side is f32.  Two passes (x, then y)
// give the gradient of madm_resize_bilinear with an f32 intermediate.
template <typename T, bool IN_T, bool OUT_T>
__global__ __launch_bounds__(256) void bilinear_adjoint_axis_kernel(const void* __restrict__ in_, size_t in_ld,
                                                                    void* __restrict__ out_, size_t out_ld, int outer,
                                                                    int Lout, int Lin, int inner) {
    constexpr int EPC = TT<T>::EPC;
    const unsigned G = (unsigned)inner / EPC;
    const size_t total = (size_t)outer * Lin * G;
    const float scale = (float)Lin / (float)Lout;
    const float inv = (float)Lout / (float)Lin;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const unsigned g = (unsigned)(idx % G);
        const size_t t = idx / G;
        const int i = (int)(t % (size_t)Lin);
        const size_t o = t / (size_t)Lin;
        int lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
        int hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
        if (lo < 0) lo = 0;
        if (hi > Lout - 1) hi = Lout - 1;
        float acc[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) acc[j] = 0.f;
        for (int l = lo; l <= hi; ++l) {
            int i0, i1;
            float l1;
            bilinear_coord(l, Lin, scale, i0, i1, l1);
            const float w = (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
            if (w == 0.f) continue;
            float f[EPC];
            const size_t off = (o * Lout + l) * in_ld + (size_t)g * EPC;
            if constexpr (IN_T) {
                chunk_to_f32<T>(*reinterpret_cast<const uint4*>((const T*)in_ + off), f);
            } else {
                const float* p = (const float*)in_ + off;
#pragma unroll
                for (int j = 0; j < EPC; j += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(p + j);
                    f[j] = v.x; f[j + 1] = v.y; f[j + 2] = v.z; f[j + 3] = v.w;
                }
            }
#pragma unroll
            for (int j = 0; j < EPC; ++j) acc[j] += w * f[j];
        }
        const size_t ooff = (o * Lin + i) * out_ld + (size_t)g * EPC;
        if constexpr (OUT_T) {
            *reinterpret_cast<uint4*>((T*)out_ + ooff) = f32_to_chunk<T>(acc);
        } else {
            float* p = (float*)out_ + ooff;
#pragma unroll
            for (int j = 0; j < EPC; j += 4) *reinterpret_cast<float4*>(p + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
        }
    }
}

// pixel-weighted softmax cross entropy with ignore index on f32 logit tokens [M][ldx] (K classes), one thread per pixel:
//   loss_i = w_i * (logsumexp(x_i) - x_i[label_i])   (0 where label == ignore);  loss_sum += sum_i loss_i  (f64)
//   dlogits[i][k] = coef * g * w_i * (softmax_k - [k == label_i])  (0 where ignored; columns K .. ldd-1 zero)
// CmdiseCriterion.cross_entropy (modeling/criterion.py:120-131): F.cross_entropy(reduction='none', ignore_index) *
// pixel_weight, then .mean() over ALL pixels -- the caller folds 1 / M and the loss weight into `coef`.
template <typename T>
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ x, int ldx, int K,
                                                         const int64_t* __restrict__ labels,
                                                         const float* __restrict__ weight, int ignore, size_t M,
                                                         double* loss_sum, const float* __restrict__ gscale, float coef,
                                                         T* __restrict__ dlogits, int ldd) {
    __shared__ double red[4];
    double local = 0.0;
    const float g = (dlogits && gscale) ? coef * gscale[0] : coef;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (size_t)gridDim.x * blockDim.x) {
        const float* xr = x + i * ldx;
        const long long lab = labels[i];
        const bool valid = lab != (long long)ignore && lab >= 0 && lab < K;
        const float w = valid ? (weight ? weight[i] : 1.f) : 0.f;
        float mx = xr[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, xr[k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(xr[k] - mx);
        if (valid && loss_sum) local += (double)(w * (logf(se) + mx - xr[lab]));
        if (dlogits) {
            T* dr = dlogits + i * ldd;
            const float gw = g * w / se;
            for (int k = 0; k < ldd; ++k) {
                float v = 0.f;
                if (k < K && valid) v = gw * expf(xr[k] - mx) - (k == (int)lab ? g * w : 0.f);
                TT<T>::st(dr + k, v);
            }
        }
    }
    if (loss_sum) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
    }
}

// masked L1 / L2 between NCHW f32 tensors pred, gt [B][C][h][w]; mask [B][Hm][Wm] is read with F.interpolate(nearest)
// to (h, w) and broadcast over C (criterion.py:236-246): loss_sum += sum |d| m  (or d^2 m);  dpred = coef g sign(d) m
// (or 2 d m)
__global__ __launch_bounds__(256) void masked_l1_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                        const float* __restrict__ mask, int B, int C, int h, int w, int Hm,
                                                        int Wm, int l2, double* loss_sum, const float* __restrict__ gscale,
                                                        float coef, float* __restrict__ dpred) {
    __shared__ double red[4];
    const size_t total = (size_t)B * C * h * w;
    const float sy = (float)Hm / (float)h, sx = (float)Wm / (float)w;
    const float g = (dpred && gscale) ? coef * gscale[0] : coef;
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w);
        const size_t t = i / w;
        const int yy = (int)(t % h);
        const size_t b = t / h / C;
        float m = 1.f;
        if (mask) {
            int my = (int)floorf((float)yy * sy), mx = (int)floorf((float)xx * sx);
            if (my > Hm - 1) my = Hm - 1;
            if (mx > Wm - 1) mx = Wm - 1;
            m = mask[(b * Hm + my) * Wm + mx];
        }
        const float d = pred[i] - gt[i];
        if (loss_sum) local += (double)((l2 ? d * d : fabsf(d)) * m);
        if (dpred) dpred[i] = g * m * (l2 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
    }
    if (loss_sum) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
    }
}

// backward of tanh_gate_kernel: g_i = sum_r dout[r][i];  dx1 += tanh(a1) g;  da1 += (1 - tanh^2(a1)) x1 g;  same for 2
__global__ void tanh_gate_bwd_kernel(const float* __restrict__ a1, const float* __restrict__ x1, const float* __restrict__ a2,
                                     const float* __restrict__ x2, const float* __restrict__ dout, float* da1, float* dx1,
                                     float* da2, float* dx2, size_t n, int repeat) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float g = 0.f;
        for (int r = 0; r < repeat; ++r) g += dout[(size_t)r * n + i];
        const float t1 = a1 ? tanhf(a1[i]) : 1.f;
        if (dx1) dx1[i] += t1 * g;
        if (da1 && a1) da1[i] += (1.f - t1 * t1) * x1[i] * g;
        if (x2) {
            const float t2 = a2 ? tanhf(a2[i]) : 1.f;
            if (dx2) dx2[i] += t2 * g;
            if (da2 && a2) da2[i] += (1.f - t2 * t2) * x2[i] * g;
        }
    }
}

// train-mode BatchNorm2d bookkeeping in one launch: per-image channel sums [B][C][2] -> batch sums st [1][C][2], and the
// running statistics r = (1 - m) r + m {mean, unbiased variance} (nn.BatchNorm2d, momentum m; NULL = skip)
__global__ void bn_fold_stats_kernel(const double* __restrict__ in, int B, int C, double count, float momentum,
                                     double* __restrict__ st, float* running_mean, float* running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < B; ++b) {
        s += in[((size_t)b * C + c) * 2];
        q += in[((size_t)b * C + c) * 2 + 1];
    }
    st[2 * c] = s;
    st[2 * c + 1] = q;
    if (running_mean) {
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
}

// ---- colour augmentation of strong_transform (utils/dacs_transforms.py:40-78 -> kornia ColorJitter / GaussianBlur2d) ----
__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }

// *out (f64, zeroed) += sum over pixels of the grey value of one RGB image [3][HW]
__global__ __launch_bounds__(256) void gray_sum_kernel(const float* __restrict__ img, size_t HW, double* out) {
    __shared__ double red[4];
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (size_t)gridDim.x * blockDim.x)
        local += (double)gray_of(img[i], img[HW + i], img[2 * HW + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// one step of kornia's ColorJitter on one RGB image [3][HW] in [0, 1] (in place allowed):
//   op 0 brightness: x * f;  1 contrast: x * f + mean_gray * (1 - f);  2 saturation: (1 - f) * gray + f * x;
//   3 hue: rgb -> hsv, h = fmod(h + f, 2 pi), hsv -> rgb.   Ops 0-2 clamp to [0, 1].
__global__ __launch_bounds__(256) void color_jitter_kernel(const float* __restrict__ in, float* __restrict__ out, size_t HW,
                                                           int op, float f, const double* __restrict__ gray_sum) {
    const float mean = (op == 1) ? (float)(gray_sum[0] / (double)HW) : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (size_t)gridDim.x * blockDim.x) {
        float r = in[i], g = in[HW + i], b = in[2 * HW + i];
        if (op == 0) {
            r = clamp01(r * f); g = clamp01(g * f); b = clamp01(b * f);
        } else if (op == 1) {
            const float m = mean * (1.f - f);
            r = clamp01(r * f + m); g = clamp01(g * f + m); b = clamp01(b * f + m);
        } else if (op == 2) {
            const float y = (1.f - f) * gray_of(r, g, b);
            r = clamp01(y + f * r); g = clamp01(y + f * g); b = clamp01(y + f * b);
        } else {
            const float TWO_PI = 6.283185307179586f;
            const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
            float dc = mx - mn;
            const float v = mx, s = dc / (mx + 1e-8f);
            if (dc == 0.f) dc = 1.f;
            const float rc = mx - r, gc = mx - g, bc = mx - b;
            float h = (r == mx) ? (bc - gc) : ((g == mx) ? (rc - bc) + 2.f * dc : (gc - rc) + 4.f * dc);   // first maximum wins
            h = h / dc / 6.f;
            h = h - floorf(h);                                   // python-style % 1.0
            h = fmodf(TWO_PI * h + f, TWO_PI);                   // torch.fmod keeps the sign of the dividend
            const float h6 = h / TWO_PI * 6.f;
            float hi = floorf(h6);
            hi = hi - 6.f * floorf(hi / 6.f);                    // % 6
            float ff = h6 - 6.f * floorf(h6 / 6.f) - hi;
            const float p = v * (1.f - s), q = v * (1.f - ff * s), t = v * (1.f - (1.f - ff) * s);
            const int k = (int)hi;
            r = (k == 0 || k == 5) ? v : ((k == 1) ? q : ((k == 4) ? t : p));
            g = (k == 1 || k == 2) ? v : ((k == 0) ? t : ((k == 3) ? q : p));
            b = (k == 3 || k == 4) ? v : ((k == 2) ? t : ((k == 5) ? q : p));
        }
        out[i] = r; out[HW + i] = g; out[2 * HW + i] = b;
    }
}

// 1-D correlation along x (axis 1) or y (axis 0) of `planes` f32 planes [H][W] with `ks` weights, reflect border
// (kornia filter2d_separable(border_type='reflect')).
__global__ __launch_bounds__(256) void blur_axis_kernel(const float* __restrict__ in, float* __restrict__ out, int planes,
                                                        int H, int W, int axis, int ks, const float* __restrict__ wts) {
    const size_t total = (size_t)planes * H * W;
    const int half = ks / 2, L = axis ? W : H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const size_t t = i / W;
        const int y = (int)(t % H);
        const float* pl = in + (t / H) * (size_t)H * W;
        const int c0 = axis ? x : y;
        float acc = 0.f;
        for (int k = 0; k < ks; ++k) {
            int c = c0 + k - half;
            if (c < 0) c = -c;
            if (c >= L) c = 2 * (L - 1) - c;
            acc += wts[k] * (axis ? pl[(size_t)y * W + c] : pl[(size_t)c * W + x]);
        }
        out[i] = acc;
    }
}

unsigned grid_for(size_t n, unsigned cap = 8192) {
    size_t g = (n + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" {

int madm_scale_channels(int dtype, const void* x, int ldx, const float* scale, void* y, int ldy, int B, int HW, int C,
                        void* stream) {
    MADM_REQUIRE(x && scale && y && B > 0 && HW > 0 && C > 0, "scale_channels: bad argument");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0 && ldx % epc == 0 && ldy % epc == 0 && ldx >= C && ldy >= C,
                 "scale_channels: C / ld must be multiples of %d elements", epc);
    const size_t total = (size_t)B * HW * (C / epc);
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (scale_channels_kernel<T><<<grid_for(total), 256, 0, s>>>((const T*)x, ldx, scale, (T*)y, ldy,
                                                                                       HW, C, total)));
    return madm_check_launch("scale_channels_kernel");
}

int madm_bn_fold_stats(const double* chsums, int B, int C, double count, float momentum, double* st, float* running_mean,
                       float* running_var, void* stream) {
    MADM_REQUIRE(chsums && st && B > 0 && C > 0 && count > 0 && (!running_mean == !running_var), "bn_fold_stats: bad argument");
    bn_fold_stats_kernel<<<(C + 255) / 256, 256, 0, (hipStream_t)stream>>>(chsums, B, C, count, momentum, st, running_mean,
                                                                          running_var);
    return madm_check_launch("bn_fold_stats_kernel");
}

int madm_relu_bwd(int dtype, const void* y, const void* dy, void* dx, size_t n, void* stream) {
    MADM_REQUIRE(y && dy && dx && n > 0, "relu_bwd: bad argument");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(n % epc == 0, "relu_bwd: n must be a multiple of %d elements", epc);
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (relu_bwd_kernel<T><<<grid_for(n / epc), 256, 0, s>>>((const T*)y, (const T*)dy, (T*)dx,
                                                                                   n / epc)));
    return madm_check_launch("relu_bwd_kernel");
}

int madm_dwconv3x3_wgrad(int dtype, const void* x, const void* dy, int lddy, float* dw, int B, int H, int W, int C,
                         int dilation, void* stream) {
    MADM_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0 && C > 0 && dilation > 0, "dwconv3x3_wgrad: bad argument");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0 && lddy % epc == 0 && lddy >= C, "dwconv3x3_wgrad: C / lddy must be multiples of %d", epc);
    hipStream_t s = (hipStream_t)stream;
    {   // large maps with whole 128-byte channel slabs: the lattice kernel (MADM_DWCONV_KERNEL=4 forces it, =1 forbids it)
        const char* fe = getenv("MADM_DWCONV_KERNEL");
        const int force = fe ? atoi(fe) : 0;
        const int chs = 8 * epc;
        const int shmax = (H + dilation - 1) / dilation, swmax = (W + dilation - 1) / dilation;
        const bool fits = C % chs == 0 && dilation > 1;
        const int ny = (shmax + 14) / 15, nx = (swmax + 14) / 15;
        // a workgroup walks the ny * nx tiles of its class one after the other (load, barrier, multiply): with few tiles
        // per class nothing covers the loads -- head tensor, 2 x 512 x 512 x 1024: 1 415 -> 709 us at dilation 6 (36
        // tiles), 1 355 -> 905 at 12 (9), 1 388 -> 1 461 at 18 (4)
        const bool pays = (size_t)B * H * W * (C / epc) >= ((size_t)1 << 21) && ny * nx >= 9;
        if (fits && (force == 4 || (force == 0 && pays))) {
            const int th = (shmax + ny - 1) / ny, tw = (swmax + nx - 1) / nx;
            const size_t blocks = (size_t)B * dilation * dilation * (C / chs);
            MADM_REQUIRE(blocks < 0x7fffffffull, "dwconv3x3_wgrad: grid too large");
            MADM_DISPATCH_DTYPE(dtype, (dwconv3x3_wgrad_lattice_kernel<T><<<(unsigned)blocks, 256, 0, s>>>(
                                           (const T*)x, (const T*)dy, lddy, dw, B, H, W, C, dilation, ny, nx, th, tw)));
            return madm_check_launch("dwconv3x3_wgrad_lattice_kernel");
        }
    }
    const size_t shm = (size_t)9 * C * sizeof(float);
    MADM_REQUIRE(shm <= 64 * 1024, "dwconv3x3_wgrad: C = %d too large", C);
    const size_t npix = (size_t)B * H * W;
    int ppb = (int)((npix + 2047) / 2048);
    if (ppb < 64) ppb = 64;
    const unsigned blocks = (unsigned)((npix + ppb - 1) / ppb);
    MADM_DISPATCH_DTYPE(dtype, (dwconv3x3_wgrad_kernel<T><<<blocks, 256, shm, s>>>((const T*)x, (const T*)dy, lddy, dw, B, H,
                                                                                 W, C, dilation, ppb)));
    return madm_check_launch("dwconv3x3_wgrad_kernel");
}

size_t madm_resize_bilinear_bwd_workspace_bytes(int B, int IW, int OH, int C) {
    if (B <= 0 || IW <= 0 || OH <= 0 || C <= 0) return 0;
    return (size_t)B * OH * IW * C * sizeof(float);
}

int madm_resize_bilinear_bwd(int dtype, const void* dout, int lddo, void* din, int B, int IH, int IW, int OH, int OW, int C,
                             void* workspace, size_t workspace_bytes, void* stream) {
    MADM_REQUIRE(dout && din && workspace && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0,
                 "resize_bilinear_bwd: bad argument");
    const int epc = madm_epc(dtype);
    MADM_REQUIRE(C % epc == 0 && lddo % epc == 0 && lddo >= C, "resize_bilinear_bwd: C / lddo must be multiples of %d", epc);
    MADM_REQUIRE(workspace_bytes >= madm_resize_bilinear_bwd_workspace_bytes(B, IW, OH, C),
                 "resize_bilinear_bwd: workspace of %zu bytes needed", madm_resize_bilinear_bwd_workspace_bytes(B, IW, OH, C));
    hipStream_t s = (hipStream_t)stream;
    const size_t t1 = (size_t)B * OH * IW * (C / epc), t2 = (size_t)B * IH * IW * (C / epc);
    // x axis: dout [B*OH][OW][C] (row stride lddo) -> tmp f32 [B*OH][IW][C]
    MADM_DISPATCH_DTYPE(dtype, (bilinear_adjoint_axis_kernel<T, true, false><<<grid_for(t1), 256, 0, s>>>(
                                   dout, (size_t)lddo, workspace, (size_t)C, B * OH, OW, IW, C)));
    // y axis: tmp [B][OH][IW*C] -> din [B][IH][IW*C]
    MADM_DISPATCH_DTYPE(dtype, (bilinear_adjoint_axis_kernel<T, false, true><<<grid_for(t2), 256, 0, s>>>(
                                   workspace, (size_t)IW * C, din, (size_t)IW * C, B, OH, IH, IW * C)));
    return madm_check_launch("bilinear_adjoint_axis_kernel");
}

int madm_softmax_ce(int dtype, const float* logits, int ldx, int K, const int64_t* labels, const float* weight,
                    int ignore_index, size_t M, double* loss_sum, const float* gscale, float coef, void* dlogits, int ldd,
                    void* stream) {
    MADM_REQUIRE(logits && labels && M > 0 && K > 0 && ldx >= K && (loss_sum || dlogits), "softmax_ce: bad argument");
    MADM_REQUIRE(!dlogits || ldd >= K, "softmax_ce: ldd = %d < K = %d", ldd, K);
    hipStream_t s = (hipStream_t)stream;
    MADM_DISPATCH_DTYPE(dtype, (softmax_ce_kernel<T><<<grid_for(M, 4096), 256, 0, s>>>(logits, ldx, K, labels, weight,
                                                                                     ignore_index, M, loss_sum, gscale, coef,
                                                                                     (T*)dlogits, ldd)));
    return madm_check_launch("softmax_ce_kernel");
}

int madm_masked_l1(const float* pred, const float* gt, const float* mask, int B, int C, int h, int w, int Hm, int Wm, int l2,
                   double* loss_sum, const float* gscale, float coef, float* dpred, void* stream) {
    MADM_REQUIRE(pred && gt && B > 0 && C > 0 && h > 0 && w > 0 && (!mask || (Hm > 0 && Wm > 0)) && (loss_sum || dpred),
                 "masked_l1: bad argument");
    masked_l1_kernel<<<grid_for((size_t)B * C * h * w, 1024), 256, 0, (hipStream_t)stream>>>(pred, gt, mask, B, C, h, w, Hm, Wm,
                                                                                            l2, loss_sum, gscale, coef, dpred);
    return madm_check_launch("masked_l1_kernel");
}

int madm_gray_sum(const float* img, size_t HW, double* out, void* stream) {
    MADM_REQUIRE(img && out && HW > 0, "gray_sum: bad argument");
    gray_sum_kernel<<<grid_for(HW, 256), 256, 0, (hipStream_t)stream>>>(img, HW, out);
    return madm_check_launch("gray_sum_kernel");
}

int madm_color_jitter_step(const float* in, float* out, size_t HW, int op, float factor, const double* gray_sum, void* stream) {
    MADM_REQUIRE(in && out && HW > 0 && op >= 0 && op <= 3 && (op != 1 || gray_sum), "color_jitter_step: bad argument");
    color_jitter_kernel<<<grid_for(HW, 2048), 256, 0, (hipStream_t)stream>>>(in, out, HW, op, factor, gray_sum);
    return madm_check_launch("color_jitter_kernel");
}

int madm_blur_axis_f32(const float* in, float* out, int planes, int H, int W, int axis, int ksize, const float* weights,
                       void* stream) {
    MADM_REQUIRE(in && out && weights && in != out && planes > 0 && H > 0 && W > 0 && (axis == 0 || axis == 1) && ksize > 0 &&
                     ksize / 2 < (axis ? W : H), "blur_axis: bad argument");
    blur_axis_kernel<<<grid_for((size_t)planes * H * W), 256, 0, (hipStream_t)stream>>>(in, out, planes, H, W, axis, ksize, weights);
    return madm_check_launch("blur_axis_kernel");
}

int madm_tanh_gate_bwd(const float* a1, const float* x1, const float* a2, const float* x2, const float* dout, float* da1,
                       float* dx1, float* da2, float* dx2, size_t n, int repeat, void* stream) {
    MADM_REQUIRE(x1 && dout && n > 0 && repeat > 0 && (!a2 || x2), "tanh_gate_bwd: bad argument");
    tanh_gate_bwd_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(a1, x1, a2, x2, dout, da1, dx1, da2, dx2, n, repeat);
    return madm_check_launch("tanh_gate_bwd_kernel");
}

}  // extern "C"
