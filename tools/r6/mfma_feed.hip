// round 6: what does a v_mfma_f32_32x32x16_bf16 cost inside a product loop?  One product wave per SIMD (256 threads), 8 accumulators,
// 48 MFMAs per step as in dense_xfwd; variants: operands constant in registers | six fragment sets rotating in registers
// (zeros / random data) | fragments read from LDS every step (14 ds_read_b128 per step) | the same beside four waves of vector work.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_feed tools/r6/mfma_feed.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: one (A, B) pair for every MFMA; 1: A[2][4], B[6][2] in registers, loaded once from `src` (zeros or random); 2: read from LDS
// every step; 3: mode 2 + waves 4..7 run a dependent vector chain (512 threads)
template <int MODE, bool AGPR> __global__ void __launch_bounds__(512) k(const u32x4* src, float* out, unsigned long long* cyc, int n) {
    extern __shared__ unsigned char lds[];
    if (AGPR) { float z = 0.0f; asm volatile("; an AGPR operand: the compiler selects the AGPR forms of the MFMAs" :: "a"(z)); }
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (unsigned i = tid; i < 65536 / 16; i += blockDim.x) reinterpret_cast<u32x4*>(lds)[i] = src[i];
    __syncthreads();
    if (wave >= 4) {
        if (MODE == 3) {
            float a = tid * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;
            for (int i = 0; i < n * 150; ++i) { a = a * b + c; d = d * b + a; c = c * 0.999f + d; b = b * 0.9999f + 1e-6f; }
            out[blockIdx.x * 512 + tid] = a + d;
        }
        return;
    }
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    bf16x8 A[2][4], B[6][2];
    const unsigned char* base = lds + lane * 16;
    for (int kc = 0; kc < 2; ++kc) for (int i = 0; i < 4; ++i) A[kc][i] = *reinterpret_cast<const bf16x8*>(base + (kc * 4 + i) * 1024);
    for (int q = 0; q < 6; ++q) for (int j = 0; j < 2; ++j) B[q][j] = *reinterpret_cast<const bf16x8*>(base + 8192 + (q * 2 + j) * 1024);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < n; ++s) {
        if (MODE >= 2) {
            const unsigned char* b2 = base + (s & 1) * 16384 + wave * 32768 / 4;
            for (int kc = 0; kc < 2; ++kc) for (int i = 0; i < 4; ++i) A[kc][i] = *reinterpret_cast<const bf16x8*>(b2 + (kc * 4 + i) * 1024);
            for (int q = 0; q < 6; ++q) for (int j = 0; j < 2; ++j) B[q][j] = *reinterpret_cast<const bf16x8*>(b2 + 8192 + ((q * 2 + j) & 7) * 1024);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MODE == 0 ? A[0][0] : A[q / 3][i], MODE == 0 ? B[0][0] : B[q][j], acc[i][j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.0f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 512 + tid] = sum;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

int main() {
    const int n = 4000;
    std::vector<unsigned> h(65536 / 4);
    u32x4* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 65536); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const void* fns[8] = {(const void*)k<0, false>, (const void*)k<1, false>, (const void*)k<2, false>, (const void*)k<3, false>, (const void*)k<0, true>, (const void*)k<1, true>, (const void*)k<2, true>, (const void*)k<3, true>};
    for (int i = 0; i < 8; ++i) hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int data = 0; data < 3; data += 2) {
        // 0: zeros, 1: small integers (few bits set), 2: random bf16 in [-2, 2)
        for (size_t i = 0; i < h.size(); ++i) {
            if (data == 0) h[i] = 0;
            else if (data == 1) h[i] = 0x40004000u + ((rand() & 3) << 23) + ((rand() & 3) << 7);
            else { unsigned a = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15) + ((rand() & 1) << 7), b = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15) + ((rand() & 1) << 7); h[i] = a | (b << 16); }
        }
        hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
        for (int blocks : {1, 256}) for (int agpr = 0; agpr < 2; ++agpr) for (int mode = 0; mode < 4; ++mode) {
            float ms = 0;
            unsigned long long c[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                const int threads = mode == 3 ? 512 : 256;
                if (mode == 0 && !agpr) k<0, false><<<blocks, threads, 65536>>>(src, out, cyc, n);
                if (mode == 0 && agpr) k<0, true><<<blocks, threads, 65536>>>(src, out, cyc, n);
                if (mode == 1 && !agpr) k<1, false><<<blocks, threads, 65536>>>(src, out, cyc, n);
                if (mode == 1 && agpr) k<1, true><<<blocks, threads, 65536>>>(src, out, cyc, n);
                if (mode == 2 && !agpr) k<2, false><<<blocks, threads, 65536>>>(src, out, cyc, n);
                if (mode == 2 && agpr) k<2, true><<<blocks, threads, 65536>>>(src, out, cyc, n);
                if (mode == 3 && !agpr) k<3, false><<<blocks, threads, 65536>>>(src, out, cyc, n);
                if (mode == 3 && agpr) k<3, true><<<blocks, threads, 65536>>>(src, out, cyc, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(c, cyc, 32, hipMemcpyDeviceToHost);
            printf("data %d  %3d workgroups  accumulators in %s  mode %d: %.1f cycles per MFMA (wave 0's counter), %.2f ns per MFMA per wave (whole launch), %.0f TFLOP/s\n", data, blocks, agpr ? "AGPRs" : "VGPRs", mode,
                   (double)c[0] / (n * 48.0), ms * 1e6 / (n * 48.0), 2.0 * 32 * 32 * 16 * n * 48.0 * 4 * blocks / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
