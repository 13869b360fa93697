// Error reporting and ABI version of libmadm_hip.
#include "common.hpp"
#include <cstdarg>
#include <cstdio>

namespace {
thread_local char g_err[512] = "";
}

void madm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int madm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        madm_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MADM_ERR_LAUNCH;
    }
    return MADM_OK;
}

namespace {
// Calibration loop of bench.py ("calib" in its JSON line): nothing but independent v_mfma_f32_16x16x32_f16, operands in
// registers, 8 accumulators per wave.  Two workgroups of four waves per CU (2 waves per SIMD, the residency of the conv
// kernels) hold the matrix pipes busy; its rate is what THIS device sustains under a chip-wide MFMA load (devices of one
// pool differ by up to 12 % there, MI355X_MICROARCH.md "DVFS give-back" item 5).
typedef float calib_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 calib_f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void calib_mfma_loop_kernel(int iters, float* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
    calib_f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = calib_f32x4{0.f, 0.f, 0.f, 0.f};
    calib_f16x8 a, b;
    // varied operands: the clock the chip holds depends on the data (zeros read up to 20 % high)
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u;
        a[i] = (_Float16)(((int)(h >> 20) - 2048) * (1.0f / 2048.f));
        h = h * 1664525u + 1013904223u;
        b[i] = (_Float16)(((int)(h >> 20) - 2048) * (1.0f / 2048.f));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
#endif
}
// Test infrastructure (tools/exp/stress_eval_f32.py, tests/test_ops_gpu.py::test_kernels_do_not_read_unwritten_lds): every CU's
// LDS filled with a pattern (quiet NaNs by default) by workgroups that each take 80 KB (two per CU = all 160 KB).  A kernel that reads LDS it has not
// written then computes garbage DETERMINISTICALLY instead of depending on what ran on the CU before it.
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned poison_smem[];
    for (int i = threadIdx.x; i < 20480; i += 256) poison_smem[i] = pattern;
    __syncthreads();
    // keep the stores alive; hold the CU for a moment so that the grid spreads over all CUs
    unsigned acc = 0;
    for (int r = 0; r < 64; ++r) acc += poison_smem[(threadIdx.x * 61 + r * 257) % 20480];
    if (acc == 0x12345678u) sink[0] = acc;
}
}  // namespace

extern "C" {
int madm_debug_poison_lds(unsigned pattern, void* sink, void* stream) {
    MADM_REQUIRE(sink != nullptr, "poison_lds: a 4-byte device sink");
    static std::atomic<uint64_t> attr_done{0};
    if (int e = madm_raise_dynamic_lds(reinterpret_cast<const void*>(poison_lds_kernel), (size_t)81920, attr_done, "poison_lds")) return e;
    poison_lds_kernel<<<dim3(2048), 256, 81920, (hipStream_t)stream>>>(pattern, (unsigned*)sink);
    return madm_check_launch("poison_lds_kernel");
}

int madm_abi_version(void) { return MADM_ABI_VERSION; }
const char* madm_last_error(void) { return g_err; }

int madm_calib_mfma_loop(int iters, int blocks, float* sink, double* flop, void* stream) {
    MADM_REQUIRE(iters > 0 && blocks > 0 && sink != nullptr, "calib: iters, blocks > 0 and a 4-byte device sink");
    calib_mfma_loop_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(iters, sink);
    if (flop) *flop = 2.0 * 16 * 16 * 32 * 8.0 * 4.0 * (double)iters * (double)blocks;   // 8 MFMAs x 4 waves per iteration
    return madm_check_launch("calib_mfma_loop_kernel");
}
}
