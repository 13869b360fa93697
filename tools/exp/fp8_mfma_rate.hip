// Matrix-pipe rate of the block-scaled e4m3 MFMA against the f16 MFMA the attention kernel uses:
// chip-wide loops of independent MFMAs, operands in registers, two workgroups of four waves per CU (2 waves per SIMD).
//   v_mfma_f32_16x16x32_f16            : 16 x 16 x 32  x 2 = 16 384 FLOP
//   v_mfma_scale_f32_16x16x128_f8f6f4  : 16 x 16 x 128 x 2 = 65 536 FLOP (e4m3 x e4m3, E8M0 scale per 32 elements)
// and a PV-shaped mix: per 128 keys of a d = 40 head (48-column tile) the f16 kernel issues 3 x 4 = 12 16x16x32 MFMAs per 16
// query rows for P.V; the scaled form needs 3 x 1.  build: hipcc -O2 --offload-arch=gfx950 fp8_mfma_rate.hip -o fp8_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void loop_f16(int iters, float* sink) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u; a[i] = (_Float16)(((int)(h >> 20) - 2048) * (1.0f / 2048.f));
        h = h * 1664525u + 1013904223u; b[i] = (_Float16)(((int)(h >> 20) - 2048) * (1.0f / 2048.f));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void loop_fp8(int iters, float* sink) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    i32x8 a, b;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) {   // random e4m3 bytes with the exponent kept small (no NaN pattern 0x7f / 0xff)
        h = h * 1664525u + 1013904223u; a[i] = (int)(h & 0x3f3f3f3fu);
        h = h * 1664525u + 1013904223u; b[i] = (int)(h & 0x3f3f3f3fu);
    }
    const int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;   // E8M0 scale 2^0 in every byte
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, sa, 0, sb);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
}

// One 128-key tile of the d = 40 flash kernel per wave and iteration, as instruction counts (attention.hip, DESIGN.md section
// 11.1): S = K Q^T 32 MFMAs, soft-max ~370 VALU slots of which 64 exponentials, then P.V as 24 f16 MFMAs (FP8 = false) or as 6
// block-scaled e4m3 MFMAs + the 16 conversions of P to e4m3 pairs (FP8 = true; P at a fixed scale, V assumed pre-quantised).
template <bool FP8>
__global__ __launch_bounds__(256) void tile_mix(int iters, float* sink) {
    f32x4 s_[8], o[6];
    for (int i = 0; i < 8; ++i) s_[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 6; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b; i32x8 pa, vb;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u; a[i] = (_Float16)(((int)(h >> 20) - 2048) * (1.0f / 2048.f));
        h = h * 1664525u + 1013904223u; b[i] = (_Float16)(((int)(h >> 20) - 2048) * (1.0f / 2048.f));
        h = h * 1664525u + 1013904223u; pa[i] = (int)(h & 0x3f3f3f3fu);
        h = h * 1664525u + 1013904223u; vb[i] = (int)(h & 0x3f3f3f3fu);
    }
    float m = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) s_[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, s_[i], 0, 0, 0);      // 32 MFMAs
        float p[32];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // per score: max, fma, exp (+ conversions below): the kernel's three-operation soft-max
                m = fmaxf(m, s_[i][r]);
                p[i * 4 + r] = __builtin_amdgcn_exp2f(fmaf(s_[i][r], 0.125f, -m));
            }
#pragma unroll
        for (int i = 0; i < 32; ++i) p[i] = fmaf(p[i], 0.999f, 1e-3f) * 0.5f + p[(i + 1) & 31] * 1e-4f;     // ~3 more slots per score pair
        if constexpr (FP8) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                pa[i] = __builtin_amdgcn_cvt_pk_fp8_f32(p[4 * i], p[4 * i + 1], __builtin_amdgcn_cvt_pk_fp8_f32(p[4 * i + 2], p[4 * i + 3], pa[i], true), false);
#pragma unroll
            for (int i = 0; i < 6; ++i)
                o[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pa, vb, o[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // 6
        } else {
            f16x8 pf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[i][j] = (_Float16)p[i * 8 + j];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 6; ++i) o[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, pf[r], o[i], 0, 0, 0);     // 24
        }
    }
    float s = m;
    for (int i = 0; i < 6; ++i) s += o[i][0] + o[i][1] + o[i][2] + o[i][3];
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    float* d; CK(hipMalloc(&d, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 512, iters = 40000;
    for (int k = 0; k < 2; ++k) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (k == 0) loop_f16<<<blocks, 256>>>(iters, d); else loop_fp8<<<blocks, 256>>>(iters, d);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = (double)blocks * 4 * iters * 8 * (k == 0 ? 16384.0 : 65536.0);
            printf("%s: %.3f ms, %.0f TFLOP/s, %.1f clocks per MFMA and SIMD at 2.4 GHz\n", k == 0 ? "v_mfma_f32_16x16x32_f16          " : "v_mfma_scale_f32_16x16x128_f8f6f4",
                   ms, flop / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * 2));
        }
    }
    for (int k = 0; k < 2; ++k) {
        for (int rep = 0; rep < 3; ++rep) {
            const int it2 = 4000;
            CK(hipEventRecord(e0));
            if (k == 0) tile_mix<false><<<blocks, 256>>>(it2, d); else tile_mix<true><<<blocks, 256>>>(it2, d);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("attention tile mix, P.V as %s: %.3f ms = %.0f clocks per tile and wave pair at 2.4 GHz\n",
                   k == 0 ? "24 x f16 16x16x32       " : "6 x scaled e4m3 16x16x128", ms, ms * 1e-3 * 2.4e9 / it2);
        }
    }
    return 0;
}
