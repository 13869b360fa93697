// Stand-alone reproducer attempt for the packed-FP32 observation of DESIGN.md (round 4, section 11.3).
//
// What the instrumented A-stationary kernel showed on MI355X (tools/exp/pkf32_check.py, profiles/round4_pkf32_check.txt): in
// the in-place LayerNorm  f = (x - mean) * rstd  hipcc emits, per pair of elements,
//     v_pk_add_f32 v[d:d+1], v[x:x+1], v[m-1:m]  op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]     ; x - mean, mean in the HIGH register
//     v_pk_mul_f32 v[d:d+1], v[r:r+1], v[d:d+1]  op_sel_hi:[0,1]                          ; * rstd, rstd in the LOW register
// and with two workgroups per CU (a second wave on the SIMD sitting in its MFMA loop) the LOW result of some wave
// instructions came out as x * rstd (mean not subtracted) in lanes 48..63 only, while v_sub_f32 / v_mul_f32 issued right
// behind them on the SAME registers gave the right value.  This program isolates the two instructions:
//
//   role A (even workgroups): ds_read_b128 + v_mfma_f32_16x16x32_f16 loop (the co-resident load)
//   role B (odd workgroups):  rows in LDS -> DPP row sums -> mean / rstd -> the packed pair (inline asm, exactly the forms
//                             above) AND the scalar pair on the same registers -> compare, count, record
//
//   hipcc --offload-arch=gfx950 -O3 tools/exp/pkf32_repro.hip -o /tmp/pkf32_repro && /tmp/pkf32_repro [iters] [partner 0/1] [partial-exec 0/1] [nops 0/1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned g_count[4];
__device__ unsigned g_rec[64 * 8];

__device__ __forceinline__ float row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));
    return x;
}

__global__ __launch_bounds__(256, 2) void repro(int iters, int partner, int partial, int nops, const float* __restrict__ src) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 72 KB per workgroup: two workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16 * 1024; i += 256) lds[i] = src[(blockIdx.x * 97 + i) & 0xffff];
    __syncthreads();
    if ((blockIdx.x & 1) == 0) {
        if (!partner) return;
        // ---- role A: LDS fragment reads + MFMAs, two waves per SIMD with role B's waves ----
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned base = (unsigned)(size_t)lds + (unsigned)lane * 16u;
        for (int it = 0; it < iters * 4; ++it) {
            u32x4 a, b;
            asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(a) : "v"(base + (unsigned)((it & 7) * 1024)));
            asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(b) : "v"(base + (unsigned)((it & 7) * 1024)));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678f) g_rec[511] = 1;
        return;
    }
    // ---- role B: 16 lanes per row, pieces of 4 floats; the third piece round has only lanes l16 < 8 active when `partial` ----
    const int l16 = lane & 15, rq = lane >> 4, wave = tid >> 6;
    unsigned n_lo = 0, n_hi = 0;
    for (int it = 0; it < iters; ++it) {
        const int row = (wave * 4 + rq + 16 * (it & 3)) & 63;
        f32x4 pc[3];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int pi = l16 + 16 * u;
            if (!partial || pi < 40) {
                pc[u] = *reinterpret_cast<const f32x4*>(lds + (row * 48 + pi) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { s += pc[u][e]; q += pc[u][e] * pc[u][e]; }
            }
        }
        s = row16_sum(s);
        q = row16_sum(q);
        const float inv_k = partial ? 1.0f / 160.f : 1.0f / 192.f;
        const float mean = s * inv_k;
        float var = q * inv_k - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int pi = l16 + 16 * u;
            if (!partial || pi < 40) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2 x = f32x2{pc[u][2 * h], pc[u][2 * h + 1]};
                    const f32x2 mm = f32x2{q, mean};        // mean in the HIGH register of the pair, as in the kernel
                    const f32x2 rr = f32x2{rstd, s};        // rstd in the LOW register
                    const unsigned lim = (partial && u == 2) ? 8u : 16u;
                    f32x2 d = f32x2{0.f, 0.f};
                    float g0 = 0.f, g1 = 0.f;
                    unsigned long long sv;
                    // the EXEC write (s_and_saveexec: lanes l16 < lim stay) directly in front of the packed pair, as hipcc
                    // lays the kernel's piece loop out (block entry = first packed op); NOPS > 0 puts wait states between
                    if (nops == 0)
                        asm volatile(
                            "v_cmp_gt_u32 vcc, %[lim], %[l16]\n\t"
                            "s_and_saveexec_b64 %[sv], vcc\n\t"
                            "v_pk_add_f32 %[d], %[x], %[mm] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                            "v_sub_f32 %[g0], %[x0], %[mean]\n\t"
                            "v_sub_f32 %[g1], %[x1], %[mean]\n\t"
                            "s_nop 1\n\t"
                            "v_pk_mul_f32 %[d], %[rr], %[d] op_sel_hi:[0,1]\n\t"
                            "v_mul_f32 %[g0], %[g0], %[rstd]\n\t"
                            "v_mul_f32 %[g1], %[g1], %[rstd]\n\t"
                            "s_or_b64 exec, exec, %[sv]"
                            : [d] "+&v"(d), [g0] "+&v"(g0), [g1] "+&v"(g1), [sv] "=&s"(sv)
                            : [x] "v"(x), [mm] "v"(mm), [x0] "v"(x[0]), [x1] "v"(x[1]), [mean] "v"(mean), [rr] "v"(rr), [rstd] "v"(rstd),
                              [lim] "v"(lim), [l16] "v"(l16)
                            : "vcc");
                    else
                        asm volatile(
                            "v_cmp_gt_u32 vcc, %[lim], %[l16]\n\t"
                            "s_and_saveexec_b64 %[sv], vcc\n\t"
                            "s_nop 7\n\t"
                            "s_nop 7\n\t"
                            "v_pk_add_f32 %[d], %[x], %[mm] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                            "v_sub_f32 %[g0], %[x0], %[mean]\n\t"
                            "v_sub_f32 %[g1], %[x1], %[mean]\n\t"
                            "s_nop 1\n\t"
                            "v_pk_mul_f32 %[d], %[rr], %[d] op_sel_hi:[0,1]\n\t"
                            "v_mul_f32 %[g0], %[g0], %[rstd]\n\t"
                            "v_mul_f32 %[g1], %[g1], %[rstd]\n\t"
                            "s_or_b64 exec, exec, %[sv]"
                            : [d] "+&v"(d), [g0] "+&v"(g0), [g1] "+&v"(g1), [sv] "=&s"(sv)
                            : [x] "v"(x), [mm] "v"(mm), [x0] "v"(x[0]), [x1] "v"(x[1]), [mean] "v"(mean), [rr] "v"(rr), [rstd] "v"(rstd),
                              [lim] "v"(lim), [l16] "v"(l16)
                            : "vcc");
                    // only the LOW result is compared: it is the half that failed in the kernel (the high half of this
                    // stand-alone pair differs from its scalar twin for a reason of the harness, not of the hardware)
                    if ((unsigned)l16 < lim && __builtin_bit_cast(unsigned, d[0]) != __builtin_bit_cast(unsigned, g0)) {
                        // counted in registers (one atomic per thread at the end: a storm of same-address atomics would
                        // look like a hang); only a thread's FIRST mismatch is recorded
                        if (__builtin_bit_cast(unsigned, d[0]) != __builtin_bit_cast(unsigned, g0)) ++n_lo; else ++n_hi;
                        if (n_lo + n_hi == 1) {
                            const unsigned k = atomicAdd(&g_count[3], 1u);
                            if (k < 64) {
                                unsigned* r = g_rec + k * 8;
                                r[0] = blockIdx.x; r[1] = tid; r[2] = (unsigned)u * 2 + h;
                                r[3] = __builtin_bit_cast(unsigned, x[0]); r[4] = __builtin_bit_cast(unsigned, mean);
                                r[5] = __builtin_bit_cast(unsigned, d[0]); r[6] = __builtin_bit_cast(unsigned, g0);
                                r[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
                            }
                        }
                    }
                    pc[u][2 * h] = d[0]; pc[u][2 * h + 1] = d[1];
                }
                *reinterpret_cast<f32x4*>(lds + (row * 48 + pi) * 4) = pc[u];
            }
        }
    }
    if (n_lo + n_hi) {
        atomicAdd(&g_count[0], n_lo + n_hi);
        atomicAdd(&g_count[1], n_lo);
        atomicAdd(&g_count[2], n_hi);
    }
}

static float f32(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const int partner = argc > 2 ? atoi(argv[2]) : 1;
    const int partial = argc > 3 ? atoi(argv[3]) : 1;
    const int nops = argc > 4 ? atoi(argv[4]) : 0;
    float* src;
    hipMalloc(&src, 65536 * 4);
    float* h = (float*)malloc(65536 * 4);
    unsigned s = 12345u;
    for (int i = 0; i < 65536; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 16) - 32768) * (1.0f / 4096.f) + 3.0f; }
    hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)repro, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    unsigned zero[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_count), zero, sizeof(zero));
        hipLaunchKernelGGL(repro, dim3(512), dim3(256), 72 * 1024, 0, iters, partner, partial, nops, src);
        hipError_t e = hipDeviceSynchronize();
        unsigned c[4], rec[64 * 8];
        hipMemcpyFromSymbol(c, HIP_SYMBOL(g_count), sizeof(c));
        hipMemcpyFromSymbol(rec, HIP_SYMBOL(g_rec), sizeof(rec));
        printf("rep %d (%s): iters %d partner-MFMA %d partial-EXEC %d nops %d: packed != scalar in %u pairs (low half %u, high half only %u)\n", rep,
               hipGetErrorString(e), iters, partner, partial, nops, c[0], c[1], c[2]);
        for (unsigned k = 0; k < c[3] && k < 6; ++k) {
            const unsigned* r = rec + k * 8;
            printf("   wg %u tid %u (lane %u) pair %u: x %.5f mean %.5f packed %.6f scalar %.6f  hw_id %08x (wave slot %u simd %u cu %u)\n", r[0], r[1],
                   r[1] & 63, r[2], f32(r[3]), f32(r[4]), f32(r[5]), f32(r[6]), r[7], r[7] & 15, (r[7] >> 4) & 3, (r[7] >> 8) & 15);
        }
    }
    return 0;
}
