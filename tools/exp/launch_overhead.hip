// Per-kernel boundary cost of dependent tiny kernels: eager stream launches vs hipGraph replay (MI355X).
// build: hipcc -O2 --offload-arch=gfx950 launch_overhead.hip -o launch_overhead
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void medium(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    float* d; CK(hipMalloc(&d, 64 << 20)); CK(hipMemset(d, 0, 64 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        const int blocks = mode == 0 ? 1 : 256 * 4;   // tiny, or one that fills the chip (1M elements)
        // eager
        for (int i = 0; i < 100; ++i) { if (mode == 0) tiny<<<1, 64, 0, s>>>(d); else medium<<<blocks, 256, 0, s>>>(d, blocks * 256); }
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) { if (mode == 0) tiny<<<1, 64, 0, s>>>(d); else medium<<<blocks, 256, 0, s>>>(d, blocks * 256); }
        auto t1 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d eager : %.2f us/kernel on the GPU timeline, host enqueue %.2f us/kernel\n", mode, ms * 1e3 / N,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) { if (mode == 0) tiny<<<1, 64, 0, s>>>(d); else medium<<<blocks, 256, 0, s>>>(d, blocks * 256); }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d graph : %.2f us/kernel\n", mode, ms * 1e3 / N);
    }
    return 0;
}
