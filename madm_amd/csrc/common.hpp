// Shared device/host helpers of libmadm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "madm_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// elements per 16-byte chunk / bytes per element of a madm_dtype (host side)
static inline int madm_epc(int dtype) { return dtype == MADM_F32 ? 4 : 8; }
static inline int madm_esize(int dtype) { return dtype == MADM_F32 ? 4 : 2; }
static inline bool madm_dtype_ok(int dtype) { return dtype == MADM_F32 || dtype == MADM_BF16 || dtype == MADM_F16; }

void madm_set_error(const char* fmt, ...);
int madm_check_launch(const char* what);

#define MADM_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            madm_set_error(__VA_ARGS__);        \
            return MADM_ERR_INVALID_ARG;        \
        }                                       \
    } while (0)

// ---- per-dtype traits: one 16-byte "chunk" holds EPC elements -------------------------
template <typename T> struct TT;
template <> struct TT<float> {
    static constexpr int EPC = 4;
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct TT<bf16_t> {
    static constexpr int EPC = 8;
    typedef bf16x8 vec8;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return (float)*p; }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)v; }
};
template <> struct TT<f16_t> {
    static constexpr int EPC = 8;
    typedef f16x8 vec8;
    static __device__ __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    static __device__ __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }
};

// unpack a 16-byte chunk into floats / pack floats into a chunk
template <typename T> __device__ __forceinline__ void chunk_to_f32(const uint4& c, float* f);
template <> __device__ __forceinline__ void chunk_to_f32<float>(const uint4& c, float* f) {
    float4 v = __builtin_bit_cast(float4, c);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <> __device__ __forceinline__ void chunk_to_f32<bf16_t>(const uint4& c, float* f) {
    bf16x8 v = __builtin_bit_cast(bf16x8, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <> __device__ __forceinline__ void chunk_to_f32<f16_t>(const uint4& c, float* f) {
    f16x8 v = __builtin_bit_cast(f16x8, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ uint4 f32_to_chunk(const float* f);
template <> __device__ __forceinline__ uint4 f32_to_chunk<float>(const float* f) {
    float4 v = make_float4(f[0], f[1], f[2], f[3]);
    return __builtin_bit_cast(uint4, v);
}
template <> __device__ __forceinline__ uint4 f32_to_chunk<bf16_t>(const float* f) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)f[i];
    return __builtin_bit_cast(uint4, v);
}
template <> __device__ __forceinline__ uint4 f32_to_chunk<f16_t>(const float* f) {
    f16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (f16_t)f[i];
    return __builtin_bit_cast(uint4, v);
}

// store 4 consecutive elements converted from f32 (8 B for bf16, 16 B for f32)
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f32x4 v) {
    bf16x4 o;
    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, f32x4 v) {
    f16x4 o;
    o[0] = (f16_t)v[0]; o[1] = (f16_t)v[1]; o[2] = (f16_t)v[2]; o[3] = (f16_t)v[3];
    *reinterpret_cast<f16x4*>(p) = o;
}
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) {
    float4 v = *reinterpret_cast<const float4*>(p);
    return f32x4{v.x, v.y, v.z, v.w};
}
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <> __device__ __forceinline__ f32x4 load4<f16_t>(const f16_t* p) {
    f16x4 v = *reinterpret_cast<const f16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void store2(T* p, float a, float b);
template <> __device__ __forceinline__ void store2<float>(float* p, float a, float b) {
    *reinterpret_cast<float2*>(p) = make_float2(a, b);
}
template <> __device__ __forceinline__ void store2<bf16_t>(bf16_t* p, float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 o;
    o[0] = (bf16_t)a; o[1] = (bf16_t)b;
    *reinterpret_cast<bf16x2*>(p) = o;
}
template <> __device__ __forceinline__ void store2<f16_t>(f16_t* p, float a, float b) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    f16x2 o;
    o[0] = (f16_t)a; o[1] = (f16_t)b;
    *reinterpret_cast<f16x2*>(p) = o;
}

// ---- one 16-byte operand chunk per lane -> MFMA ----------------------------------------
// Both operands are read as "row (lane & 15), 16-byte chunk (lane >> 4)" of a [rows][K] tile.
// bf16 / f16: one v_mfma_f32_16x16x32_bf16 / _f16 (lane group g holds k = 8g .. 8g+7).
// f32 : four v_mfma_f32_16x16x4_f32, element s of every lane's float4 at step s -- a
//       permutation of k that is the same for both operands, so the sum is unchanged.
// D[i][j] = sum_k a(row i, k) * b(row j, k); lane holds j = lane & 15, i = 4*(lane>>4) + reg.
template <typename T>
__device__ __forceinline__ void mma16(const uint4& a, const uint4& b, f32x4& c);
template <>
__device__ __forceinline__ void mma16<bf16_t>(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16<f16_t>(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                               __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16<float>(const uint4& a, const uint4& b, f32x4& c) {
    float4 x = __builtin_bit_cast(float4, a), y = __builtin_bit_cast(float4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, y.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, y.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, y.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, y.w, c, 0, 0, 0);
}

// x * sigmoid(x); v_exp_f32 + v_rcp_f32 (1 ulp) instead of an IEEE divide: this runs per staged element inside the
// GroupNorm-fused conv, where the divide sequence alone cost more VALU issue than the tile's MFMAs
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// madm_act: 0 = none, 1 = SiLU, 2 = ReLU
__device__ __forceinline__ float act_f(float x, int act) {
    return act == 1 ? silu_f(x) : (act == 2 ? fmaxf(x, 0.f) : x);
}
// the same over a register array with the (wave-uniform) selector tested once, not per element
template <int N>
__device__ __forceinline__ void act_inplace(float (&f)[N], int act) {
    if (act == 1) {
#pragma unroll
        for (int j = 0; j < N; ++j) f[j] = silu_f(f[j]);
    } else if (act == 2) {
#pragma unroll
        for (int j = 0; j < N; ++j) f[j] = fmaxf(f[j], 0.f);
    }
}
// exact-erf GELU (diffusers GEGLU): 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz-Stegun 7.1.26 (|error| <
// 5e-7 on erf: one v_rcp + one v_exp + 6 FMAs instead of libm's ~45-instruction erff -- the GEGLU epilogue evaluates
// 16 of them per thread and was a third of the block time of the K = 320 feed-forward GEMMs).  1 + erf is formed
// without cancellation on the negative side.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float pe = poly * __expf(-z * z);          // 1 - erf(|x| / sqrt 2)
    return 0.5f * x * (x < 0.f ? pe : 2.0f - pe);
}

// All kernel-argument cache lines requested at once, first thing in a kernel.  The compiler fetches a by-value parameter
// struct piecemeal, where each field is first used: three or four DEPENDENT scalar-cache misses (~0.3 us each on a cold
// scalar cache -- every block of a one-round launch is the first on its CU) in front of the first operand load of the
// igemm kernels (in-kernel stamps, tools/exp/stamps_reg.py: 1 400 clocks from entry to the first load).  One blocking touch
// of every 64-byte line makes the later loads hit.  LINES x 64 B must cover the explicit arguments and the hidden ones
// behind them (grid size ...).
template <int LINES>
__device__ __forceinline__ void kernarg_touch() {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(LINES >= 1 && LINES <= 6, "kernarg_touch: 1 .. 6 lines");
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    unsigned t0, t1, t2, t3, t4, t5;
    if constexpr (LINES == 6)
        asm volatile("s_load_dword %0, %6, 0x0\n\ts_load_dword %1, %6, 0x40\n\ts_load_dword %2, %6, 0x80\n\t"
                     "s_load_dword %3, %6, 0xc0\n\ts_load_dword %4, %6, 0x100\n\ts_load_dword %5, %6, 0x140\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5) : "s"(ka) : "memory");
    else if constexpr (LINES == 5)
        asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %5, 0x40\n\ts_load_dword %2, %5, 0x80\n\t"
                     "s_load_dword %3, %5, 0xc0\n\ts_load_dword %4, %5, 0x100\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4) : "s"(ka) : "memory");
    else if constexpr (LINES == 4)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\t"
                     "s_load_dword %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(ka) : "memory");
    else if constexpr (LINES == 3)
        asm volatile("s_load_dword %0, %3, 0x0\n\ts_load_dword %1, %3, 0x40\n\ts_load_dword %2, %3, 0x80\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1), "=&s"(t2) : "s"(ka) : "memory");
    else if constexpr (LINES == 2)
        asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1) : "s"(ka) : "memory");
    else
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0) : "s"(ka) : "memory");
#endif
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: set it once per device (bit per
// device id in ``done``, atomic: host threads may launch concurrently).  ``done`` is a function-local static of the caller,
// i.e. one per kernel instantiation.
#include <atomic>
inline int madm_raise_dynamic_lds(const void* kern, size_t lds, std::atomic<uint64_t>& done, const char* what) {
    if (lds <= 64 * 1024) return MADM_OK;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return MADM_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        madm_set_error("%s: cannot raise dynamic LDS to %zu: %s", what, lds, hipGetErrorString(e));
        return MADM_ERR_LAUNCH;
    }
    done.fetch_or(bit, std::memory_order_release);
    return MADM_OK;
}

#define MADM_DISPATCH_DTYPE(dtype, ...)                          \
    do {                                                         \
        if ((dtype) == MADM_F32) {                               \
            typedef float T;                                     \
            __VA_ARGS__;                                         \
        } else if ((dtype) == MADM_BF16) {                       \
            typedef bf16_t T;                                    \
            __VA_ARGS__;                                         \
        } else if ((dtype) == MADM_F16) {                        \
            typedef f16_t T;                                     \
            __VA_ARGS__;                                         \
        } else {                                                 \
            madm_set_error("unknown dtype %d", (int)(dtype));    \
            return MADM_ERR_INVALID_ARG;                         \
        }                                                        \
    } while (0)
