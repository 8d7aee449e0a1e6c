// micro-probe: per-operation latency of the primitives the interpreter is built from
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#if defined(__HIP_DEVICE_COMPILE__)
#define CAS __attribute__((address_space(4)))
#else
#define CAS
#endif
struct Insn { uint32_t w[8]; };
extern __shared__ float lds[];

__global__ void probe(const uint4* code, int n_insn, int reps, unsigned long long* out, float* sink) {
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    unsigned long long t0, t1;
    float acc = 0.f;
    // (a) dependent chain of scalar instruction fetches
    uint32_t x = 0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
        for (int pc = 0; pc < n_insn; ++pc) {
            Insn I = *((const CAS Insn*)code + pc);
            x += I.w[0] + I.w[7];
        }
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[0] = (t1 - t0) / (unsigned long long)(reps * n_insn);
    // (b) fetch + one LDS read whose address depends on the instruction
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
        for (int pc = 0; pc < n_insn; ++pc) {
            Insn I = *((const CAS Insn*)code + pc);
            acc += lds[(I.w[1] & 1023u) + tid];
        }
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[1] = (t1 - t0) / (unsigned long long)(reps * n_insn);
    // (c) fetch + 5 LDS reads + log + div + LDS write (a NAFF-like body)
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
        for (int pc = 0; pc < n_insn; ++pc) {
            Insn I = *((const CAS Insn*)code + pc);
            float A = lds[(I.w[2] & 1023u)], B = lds[(I.w[3] & 1023u) + tid], C = lds[(I.w[4] & 1023u)], S = lds[(I.w[5] & 1023u)] + 2.0f;
            float v = lds[(I.w[1] & 1023u) + tid];
            float loc = A * B + C, d = v - loc;
            acc += -(d * d) / (2.0f * S * S) - logf(S);
            lds[(I.w[1] & 1023u) + tid + 2048] = acc;
        }
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[2] = (t1 - t0) / (unsigned long long)(reps * n_insn);
    // (d) dependent LDS read chain
    uint32_t idx = tid & 63;
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps * n_insn; ++r) idx = ((uint32_t)lds[idx]) & 1023u;
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[3] = (t1 - t0) / (unsigned long long)(reps * n_insn);
    // (e) s_memtime overhead
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < 64; ++r) x += (uint32_t)__builtin_amdgcn_s_memtime();
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[4] = (t1 - t0) / 64;
    sink[tid] = acc + (float)x + (float)idx;
}
int main() {
    const int n = 63, reps = 50;
    uint32_t h[63 * 8];
    for (int i = 0; i < n * 8; ++i) h[i] = (uint32_t)(i * 2654435761u) >> 7;
    uint4* code; unsigned long long* out; float* sink;
    hipMalloc(&code, sizeof(h)); hipMalloc(&out, 64); hipMalloc(&sink, 4096 * 4);
    hipMemcpy(code, h, sizeof(h), hipMemcpyHostToDevice);
    for (int waves : {1, 5, 16}) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves), 48 * 1024, 0, code, n, reps, out, sink);
        hipDeviceSynchronize();
        unsigned long long r[5]; hipMemcpy(r, out, 40, hipMemcpyDeviceToHost);
        printf("waves=%2d  s_load x8 fetch: %llu cyc | +1 LDS read: %llu | NAFF-like body: %llu | dependent ds_read: %llu | s_memtime: %llu\n", waves, r[0], r[1], r[2], r[3], r[4]);
    }
    return 0;
}
