// Training-step tail on flat fp32 parameter storage: the per-parameter Python loops / multi-pass
// torch ops of the reference become three HBM-bound streaming kernels over ONE contiguous buffer:
//   madm_sumsq_f32     -- sum of squares in f64 (the norm of clip_grad_norm_, engine/train_loop.py:203-217)
//   madm_adamw_step    -- torch.optim.AdamW step (config_files/common/optim.py:8-17) with the GradScaler unscale and
//                         the clip coefficient folded in as one gradient scale
//   madm_ema_update    -- teacher EMA (modeling/meta_arch/cmdise.py:337-349)
// 16-byte accesses per lane, grid-stride; algorithmic traffic 28 / 12 / 4 bytes per element.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, double* __restrict__ out) {
    const size_t n4 = n / 4;
    float acc = 0.f;   // per-thread partial over at most a few thousand elements, then f64
    double dacc = 0.0;
    int cnt = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        if (++cnt == 64) { dacc += (double)acc; acc = 0.f; cnt = 0; }
    }
    dacc += (double)acc;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = n4 * 4; i < n; ++i) dacc += (double)x[i] * (double)x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = dacc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s[0] + s[1] + s[2] + s[3]);
}

struct AdamP {
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, gscale;
};

__device__ __forceinline__ void adamw1(float& p, float g, float& m, float& v, const AdamP& a) {
    g *= a.gscale;
    p *= (1.f - a.lr * a.wd);
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= (a.lr / a.bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n, AdamP a) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        adamw1(pp.x, gg.x, mm.x, vv.x, a); adamw1(pp.y, gg.y, mm.y, vv.y, a);
        adamw1(pp.z, gg.z, mm.z, vv.z, a); adamw1(pp.w, gg.w, mm.w, vv.w, a);
        reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = n4 * 4; i < n; ++i) adamw1(p[i], g[i], m[i], v[i], a);
}

// per-tensor hyper-parameters: the flat buffer is laid out in 1024-element chunks that never straddle two tensors
// (optim.FlatParams(align=1024)); chunk_tensor[c] names the tensor of chunk c and hyper[t] = {lr, weight_decay, bias
// correction 1, sqrt(bias correction 2)} -- bc1 == 0 marks a tensor that is skipped this step (no gradient arrived:
// torch.optim.AdamW leaves such parameters, their moments and their step counts untouched).
__global__ __launch_bounds__(256) void adamw_table_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, size_t nchunks,
                                                          const int* __restrict__ chunk_tensor,
                                                          const float4* __restrict__ hyper, float beta1, float beta2,
                                                          float eps, float gscale) {
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const float4 h = hyper[chunk_tensor[c]];
        if (h.z == 0.f) continue;
        AdamP a;
        a.lr = h.x; a.wd = h.y; a.bc1 = h.z; a.bc2_sqrt = h.w; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.gscale = gscale;
        const size_t i = c * 256 + threadIdx.x;
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        adamw1(pp.x, gg.x, mm.x, vv.x, a); adamw1(pp.y, gg.y, mm.y, vv.y, a);
        adamw1(pp.z, gg.z, mm.z, vv.z, a); adamw1(pp.w, gg.w, mm.w, vv.w, a);
        reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, size_t n,
                                                  float alpha) {
    const size_t n4 = n / 4;
    const float b = 1.f - alpha;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 e = reinterpret_cast<float4*>(ema)[i];
        const float4 q = reinterpret_cast<const float4*>(p)[i];
        e.x = alpha * e.x + b * q.x; e.y = alpha * e.y + b * q.y; e.z = alpha * e.z + b * q.z; e.w = alpha * e.w + b * q.w;
        reinterpret_cast<float4*>(ema)[i] = e;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = n4 * 4; i < n; ++i) ema[i] = alpha * ema[i] + b * p[i];
}

unsigned grid_for(size_t n4) {
    size_t g = (n4 + 255) / 256;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" {

int madm_sumsq_f32(const float* x, size_t n, double* out, void* stream) {
    MADM_REQUIRE(x && out && n > 0 && ((uintptr_t)x % 16) == 0, "sumsq: bad args (x must be 16-byte aligned)");
    sumsq_kernel<<<grid_for(n / 4), 256, 0, (hipStream_t)stream>>>(x, n, out);
    return madm_check_launch("sumsq_kernel");
}

int madm_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, float grad_scale, void* stream) {
    MADM_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw_step: bad args");
    MADM_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
                 "adamw_step: buffers must be 16-byte aligned");
    AdamP a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.gscale = grad_scale;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    adamw_kernel<<<grid_for(n / 4), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, a);
    return madm_check_launch("adamw_kernel");
}

int madm_adamw_step_table(float* p, const float* g, float* m, float* v, size_t n, const int* chunk_tensor, const float* hyper,
                          float beta1, float beta2, float eps, float grad_scale, void* stream) {
    MADM_REQUIRE(p && g && m && v && chunk_tensor && hyper && n > 0 && n % 1024 == 0, "adamw_step_table: bad args (n %% 1024)");
    MADM_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                     ((uintptr_t)hyper % 16) == 0, "adamw_step_table: buffers must be 16-byte aligned");
    const size_t nchunks = n / 1024;
    size_t grid = nchunks < 65536 ? nchunks : 65536;
    adamw_table_kernel<<<(unsigned)grid, 256, 0, (hipStream_t)stream>>>(p, g, m, v, nchunks, chunk_tensor, (const float4*)hyper,
                                                                      beta1, beta2, eps, grad_scale);
    return madm_check_launch("adamw_table_kernel");
}

int madm_ema_update(float* ema, const float* p, size_t n, float alpha, void* stream) {
    MADM_REQUIRE(ema && p && n > 0 && ((uintptr_t)ema % 16) == 0 && ((uintptr_t)p % 16) == 0, "ema_update: bad args");
    ema_kernel<<<grid_for(n / 4), 256, 0, (hipStream_t)stream>>>(ema, p, n, alpha);
    return madm_check_launch("ema_kernel");
}

}  // extern "C"
