// round 6: v_mfma_f32_32x32x16_bf16 back to back from registers against the NUMBER of accumulators in the rotation (distinct start values,
// so that the compiler cannot merge them) and v_mfma_f32_16x16x32_bf16 for comparison.  One wave per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_acc tools/r6/mfma_acc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int WIDE, int LB> __global__ void __launch_bounds__(LB) k(float* out, unsigned long long* cyc, int n) {
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f / (1 + i)); }
    float s = 0.0f;
    unsigned long long t0, t1;
    if (WIDE) {
        f32x16 acc[NACC];
        for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = a * 0.25f + r + threadIdx.x;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    } else {
        f32x4 acc[NACC];
        for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = a * 0.25f + r + threadIdx.x;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[a], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, int WIDE, int LB = 256> void run(float* out, unsigned long long* cyc, int blocks) {
    const int n = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k<NACC, WIDE, LB><<<blocks, 256>>>(out, cyc, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = WIDE ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    printf("%s, launch bound %d, %2d accumulators, %3d workgroups: %.1f cycles per MFMA (wave 0's counter), %.2f ns per MFMA per wave, %.0f TFLOP/s\n",
           WIDE ? "32x32x16" : "16x16x32", LB, NACC, blocks, (double)c / (n * 16.0), ms * 1e6 / (n * 16.0), flop * n * 16.0 * 4 * blocks / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
    for (int blocks : {1, 256}) {
        run<1, 1>(out, cyc, blocks); run<2, 1>(out, cyc, blocks); run<4, 1>(out, cyc, blocks); run<8, 1>(out, cyc, blocks); run<4, 1, 512>(out, cyc, blocks); run<8, 1, 512>(out, cyc, blocks); run<4, 1, 1024>(out, cyc, blocks);
        run<1, 0>(out, cyc, blocks); run<2, 0>(out, cyc, blocks); run<4, 0>(out, cyc, blocks); run<8, 0>(out, cyc, blocks); run<16, 0>(out, cyc, blocks);
    }
    return 0;
}
