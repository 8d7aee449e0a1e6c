// Diagnostic: the floor of an f32-input MFMA stream launched like dense_forward
// (512 workgroups x 4 waves, 2000 v_mfma_f32_16x16x4_f32 per wave on 10 independent accumulators),
// with nothing else in the kernel.  hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256, 2) probe(float* out, int iters, float a0, float b0) {
    f32x4 acc[10];
    for (int c = 0; c < 10; ++c) acc[c] = 0.0f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 10; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int c = 0; c < 10; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ((unsigned long long*)(out + 512 * 256))[0] = t1 - t0;
}
int main() {
    float* out;
    hipMalloc(&out, 512 * 256 * 4 + 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {512, 1024, 2048}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(probe, dim3(grid > 512 ? 512 : grid), dim3(256), 0, 0, out, 200 * (grid / 512), 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long cyc; hipMemcpy(&cyc, out + 512 * 256, 8, hipMemcpyDeviceToHost);
            const double mfma = 512.0 * 4 * 2000 * (grid / 512);
            printf("mfma/wave=%d: %.1f us per launch, %.1f TFLOP/s, wave cycles %llu (%.2f GHz if the wave spans the launch), %.1f cyc/MFMA/SIMD\n",
                   2000 * (grid / 512), ms * 1e3 / 20, mfma * 2048 / (ms * 1e-3 / 20) / 1e12, cyc, cyc / (ms * 1e3 / 20) / 1e3, cyc / (2000.0 * (grid / 512) * 2));
        }
    }
    return 0;
}
