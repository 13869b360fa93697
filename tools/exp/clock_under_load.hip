// What shader clock does the chip hold under a chip-filling MFMA loop / a VALU loop / an idle-ish loop?
// s_memtime (shader-clock counter) deltas of one wave against the kernel's wall time from HIP events.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/clock_under_load.hip -o /tmp/clock_under_load && /tmp/clock_under_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_loop(unsigned long long* out, int iters, float* sink) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        // inline asm: the intrinsic form made the compiler shuffle the accumulators through AGPR copies every iteration
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t0; out[1] = t1; }
}

__global__ __launch_bounds__(256) void valu_loop(unsigned long long* out, int iters, float* sink) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] * 1.0001f + 0.5f;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t0; out[1] = t1; }
}

template <typename K>
void run(const char* name, K kern, int blocks, int iters, double flop_per_iter_per_block) {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 16);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double cyc = (double)(h[1] - h[0]);
        printf("%-34s blocks %5d: %8.3f ms, %12.0f counter ticks in the loop -> %.3f GHz if they are shader clocks",
               name, blocks, ms, cyc, cyc / (ms * 1e6));
        if (flop_per_iter_per_block > 0) printf(", %.0f TFLOP/s", flop_per_iter_per_block * iters * blocks / (ms * 1e-3) / 1e12);
        printf("\n");
    }
}

int main() {
    // 2 blocks of 4 waves per CU = 2 waves per SIMD, like the conv kernels
    run("MFMA 16x16x32 f16, chip full", mfma_loop, 512, 200000, 8.0 * 16 * 16 * 32 * 2 * 4);
    run("MFMA 16x16x32 f16, one block", mfma_loop, 1, 200000, 8.0 * 16 * 16 * 32 * 2 * 4);
    run("VALU fma, chip full", valu_loop, 512, 400000, 0);
    run("VALU fma, one block", valu_loop, 1, 400000, 0);
    return 0;
}
