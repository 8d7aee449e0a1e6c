// round 5: issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950 — one wave per SIMD (256 threads, one workgroup), chains of
// dependent and of independent instructions.  build: hipcc --offload-arch=gfx950 -O3 -o pk_rate tools/r5/pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void __launch_bounds__(1024) k(float* out, long long* cycles, int n) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 1.0001f, c = 1e-4f;
    const f2 pm = {m, m}, pc = {c, c};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {          // 8 dependent v_fma_f32
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
        } else if (MODE == 1) {   // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 2) {   // 8 dependent v_pk_fma_f32
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pm), "v"(pc));
        } else {                  // 8 v_pk_fma_f32, four independent chains
#pragma unroll
            for (int u = 0; u < 2; ++u)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024); hipMalloc(&cyc, 8);
    const int n = 100000;
    const char* names[4] = {"v_fma_f32 dependent", "v_fma_f32 independent", "v_pk_fma_f32 dependent", "v_pk_fma_f32 independent x4"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int threads : {64, 256, 512, 768, 1024}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<1, threads>>>(out, cyc, n);
                if (mode == 1) k<1><<<1, threads>>>(out, cyc, n);
                if (mode == 2) k<2><<<1, threads>>>(out, cyc, n);
                if (mode == 3) k<3><<<1, threads>>>(out, cyc, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-32s %3d threads: %.2f ns per instruction per wave (%.2f cycles at 2.4 GHz)\n", names[mode], threads, ms * 1e6 / (8.0 * n), ms * 1e6 / (8.0 * n) * 2.4);
        }
    }
    return 0;
}
