// round 5: issue rate of v_mfma_f32_32x32x16_bf16 from registers — N independent accumulators per wave, W waves per SIMD, every CU
// busy or one.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/r5/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC> __global__ void __launch_bounds__(1024) k(float* out, int n) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f / (1 + i)); }
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.0f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int n = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 256}) for (int threads : {256, 512, 1024}) for (int nacc : {1, 2, 4}) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (nacc == 1) k<1><<<blocks, threads>>>(out, n);
            if (nacc == 2) k<2><<<blocks, threads>>>(out, n);
            if (nacc == 4) k<4><<<blocks, threads>>>(out, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double per = ms * 1e6 / ((double)n * nacc);         // ns per MFMA per wave
        const double tf = 2.0 * 32 * 32 * 16 * (double)n * nacc * (threads / 64) * blocks / (ms * 1e-3) / 1e12;
        printf("%3d workgroups x %4d threads, %d accumulators: %.2f ns per MFMA per wave, %.0f TFLOP/s%s\n", blocks, threads, nacc, per, tf,
               blocks == 256 ? "" : " (one CU)");
    }
    return 0;
}
