// round 5: a float4 sum over R "rank" buffers — the arithmetic of an all-reduce's reduction step, compiled WITH
// packed-f32 instructions (v_pk_add_f32: what RCCL's kernels are made of) — as a shared library the probe launches beside the
// library's bf16-MFMA products.  build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/r5/pk_victim.hip -o tools/bin/libpkvictim.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) sum_ranks(const float* __restrict__ in, float* __restrict__ out, int n4, int ranks, int passes) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int p = 0; p < passes; ++p)
        for (int r = 0; r < ranks; ++r) acc += reinterpret_cast<const f32x4*>(in)[(size_t)r * n4 + i];
    reinterpret_cast<f32x4*>(out)[i] = acc;
}

extern "C" int pk_victim_sum(const float* in, float* out, int n_floats, int ranks, int passes, void* stream) {
    const int n4 = n_floats / 4;
    hipLaunchKernelGGL(sum_ranks, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, out, n4, ranks, passes);
    return (int)hipGetLastError();
}
