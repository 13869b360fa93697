// Label / pseudo-label pipeline of the self-training step on the device: the reference
// does these on the host every step (PIL palette through .cpu(), softmax/max/threshold with .item(), ClassMix
// with python loops) -- /root/reference/modeling/meta_arch/mtmadise.py:159-175,339-349 and
// /root/reference/utils/dacs_transforms.py:81-111.  All index work is bit-exact; everything is HBM-bound byte /
// integer traffic: one pass, coalesced, no collectives.
#include "common.hpp"

namespace {

// label [B][HW] i64 -> rgb [B][3][HW] f32 = (palette[(uint8)label][c] / 255 - 0.5) / 0.5, valid [B][HW] = label != 255
// (mtmadise.py:159-175: astype(uint8) -> 'P' image -> putpalette(768 entries) -> RGB -> (x / 255 - 0.5) / 0.5)
__global__ __launch_bounds__(256) void label_to_rgb_kernel(const int64_t* __restrict__ label,
                                                           const unsigned char* __restrict__ palette,
                                                           float* __restrict__ rgb, float* __restrict__ valid, int B,
                                                           int HW) {
    __shared__ float lut[768];   // the normalised colours, built with the reference's f32 operation order
    for (int i = threadIdx.x; i < 768; i += 256) lut[i] = ((float)palette[i] / 255.0f - 0.5f) / 0.5f;
    __syncthreads();
    const size_t total = (size_t)B * HW;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t b = idx / HW, px = idx - b * HW;
        const int64_t l = label[idx];
        const unsigned k = (unsigned)(l & 255);   // numpy astype(uint8) wraps
        float* o = rgb + b * 3 * (size_t)HW + px;
        o[0] = lut[k * 3 + 0];
        o[(size_t)HW] = lut[k * 3 + 1];
        o[2 * (size_t)HW] = lut[k * 3 + 2];
        if (valid) valid[idx] = (l != 255) ? 1.0f : 0.0f;
    }
}

// logits [B][K][HW] f32 -> prob [B][HW] = max_k softmax_k, label [B][HW] = first argmax, *count += #(prob >= thr)
// (mtmadise.py:340-348: softmax(dim=1) -> max(dim=1) -> ge(threshold) -> sum().item())
__global__ __launch_bounds__(256) void pseudo_label_kernel(const float* __restrict__ logits, float* __restrict__ prob,
                                                           int64_t* __restrict__ label, unsigned long long* count,
                                                           int B, int K, int HW, float thr) {
    const size_t total = (size_t)B * HW;
    unsigned mine = 0;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t b = idx / HW, px = idx - b * HW;
        const float* x = logits + b * K * (size_t)HW + px;
        float m = x[0];
        int am = 0;
        for (int k = 1; k < K; ++k) {
            const float v = x[(size_t)k * HW];
            if (v > m) { m = v; am = k; }   // strict: the first maximum wins, like torch.max / torch.argmax
        }
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(x[(size_t)k * HW] - m);
        const float p = 1.0f / s;            // exp(0) / sum
        // the reference takes the max / argmax of the f32 PROBABILITIES (mtmadise.py:340-341): logits one ulp apart
        // round to equal probabilities and the FIRST of them wins there, so the label is the first k whose
        // exp(x_k - m) / s equals the maximum probability, not the largest logit
        if (am > 0) {
            for (int k = 0; k < am; ++k)
                if (expf(x[(size_t)k * HW] - m) / s == p) { am = k; break; }
        }
        prob[idx] = p;
        label[idx] = am;
        mine += (p >= thr) ? 1u : 0u;
    }
    // one atomic per block
    __shared__ unsigned part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0 && count) {
        const unsigned t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(count, (unsigned long long)t);
    }
}

// presence[v] = 1 for every label value v (0..255) that occurs (torch.unique without the sort)
__global__ __launch_bounds__(256) void label_presence_kernel(const int64_t* __restrict__ label, size_t n,
                                                             unsigned* __restrict__ presence) {
    __shared__ unsigned seen[256];
    seen[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        seen[(unsigned)(label[i] & 255)] = 1u;   // benign race: every writer stores 1
    __syncthreads();
    if (seen[threadIdx.x]) presence[threadIdx.x] = 1u;
}

// ClassMix (dacs_transforms.py:92-111): mask = label0 in chosen classes; out = mask * x0 + (1 - mask) * x1 for the
// C image planes (f32) and the label plane (i64); chosen[256] = 1 for the selected classes of THIS image
__global__ __launch_bounds__(256) void class_mix_kernel(const int64_t* __restrict__ label0,
                                                        const int64_t* __restrict__ label1,
                                                        const unsigned char* __restrict__ chosen,
                                                        const float* __restrict__ img0, const float* __restrict__ img1,
                                                        int C, int HW, float* __restrict__ mask_out,
                                                        float* __restrict__ img_out, int64_t* __restrict__ label_out) {
    __shared__ unsigned char sel[256];
    sel[threadIdx.x] = chosen[threadIdx.x];
    __syncthreads();
    for (size_t px = (size_t)blockIdx.x * 256 + threadIdx.x; px < (size_t)HW; px += (size_t)gridDim.x * 256) {
        const int64_t l0 = label0[px];
        const bool m = (l0 >= 0 && l0 < 256) && sel[l0];
        if (mask_out) mask_out[px] = m ? 1.0f : 0.0f;
        if (img_out)
            for (int c = 0; c < C; ++c) img_out[(size_t)c * HW + px] = m ? img0[(size_t)c * HW + px] : img1[(size_t)c * HW + px];
        if (label_out) label_out[px] = m ? l0 : label1[px];
    }
}

unsigned grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" {

int madm_label_to_rgb(const int64_t* label, const unsigned char* palette768, float* rgb, float* valid, int B, int HW,
                      void* stream) {
    MADM_REQUIRE(label && palette768 && rgb && B > 0 && HW > 0, "label_to_rgb: bad args");
    label_to_rgb_kernel<<<grid_for((size_t)B * HW), 256, 0, (hipStream_t)stream>>>(label, palette768, rgb, valid, B, HW);
    return madm_check_launch("label_to_rgb_kernel");
}

int madm_pseudo_label(const float* logits, float* prob, int64_t* label, unsigned long long* count, int B, int K, int HW,
                      float threshold, void* stream) {
    MADM_REQUIRE(logits && prob && label && B > 0 && K > 0 && HW > 0, "pseudo_label: bad args");
    pseudo_label_kernel<<<grid_for((size_t)B * HW), 256, 0, (hipStream_t)stream>>>(logits, prob, label, count, B, K, HW,
                                                                                  threshold);
    return madm_check_launch("pseudo_label_kernel");
}

int madm_label_presence(const int64_t* label, size_t n, unsigned* presence256, void* stream) {
    MADM_REQUIRE(label && presence256 && n > 0, "label_presence: bad args");
    size_t g = (n + 256 * 16 - 1) / (256 * 16);
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    label_presence_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(label, n, presence256);
    return madm_check_launch("label_presence_kernel");
}

int madm_class_mix(const int64_t* label0, const int64_t* label1, const unsigned char* chosen256, const float* img0,
                   const float* img1, int C, int HW, float* mask_out, float* img_out, int64_t* label_out, void* stream) {
    MADM_REQUIRE(label0 && chosen256 && HW > 0 && C >= 0, "class_mix: bad args");
    MADM_REQUIRE(!img_out || (img0 && img1 && C > 0), "class_mix: image mixing needs both images");
    MADM_REQUIRE(!label_out || label1, "class_mix: label mixing needs the second label map");
    class_mix_kernel<<<grid_for((size_t)HW), 256, 0, (hipStream_t)stream>>>(label0, label1, chosen256, img0, img1, C, HW,
                                                                          mask_out, img_out, label_out);
    return madm_check_launch("class_mix_kernel");
}

}  // extern "C"
