// round 5: a standalone repro — no libbsvi — of the effect profiles/r4/x6_notes.txt section 4 found inside the
// library: does a kernel executing PACKED-f32 VALU instructions (v_pk_fma_f32 / v_pk_add_f32) return different bits while a bf16-MFMA
// kernel is resident on the same CUs from another stream?
//
//   victim   outer-product accumulation like the library's outer_kernel: dW[M][N] += sum_k a[k][m] * b[k][n4] on float4 / float2
//            values (packed: the compiler forms v_pk_fma_f32; scalar control: the same sums with single v_fma_f32), a fixed order of
//            additions per thread and a fixed-order tree over the workgroups -> bit-reproducible alone.
//   partner  a register-resident loop of v_mfma_f32_32x32x16_bf16 (control: v_mfma_f32_32x32x2_f32), one 256-thread workgroup per CU,
//            launched on a second stream so that victim waves share the CUs with it.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/r5/pk_mfma_repro.hip -o tools/bin/pk_mfma_repro     run: tools/bin/pk_mfma_repro [launches]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int M = 4, N = 512, K = 25600, ROWS_PER_BLOCK = 100;      // the encoder heads of cfg 5: dW[4][512] = dY[K][4]^T h[K][512]

// one thread = 4 columns; a workgroup walks ROWS_PER_BLOCK rows; partial[block][m][n]
template <bool PACKED>
__global__ void __launch_bounds__(128) victim(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ partial) {
    const int n4 = threadIdx.x, k0 = blockIdx.x * ROWS_PER_BLOCK;
    __shared__ float as[ROWS_PER_BLOCK * M];
    for (int i = threadIdx.x; i < ROWS_PER_BLOCK * M; i += 128) as[i] = a[(size_t)k0 * M + i];
    __syncthreads();
    if (PACKED) {
        f32x2 acc[M][2];
        for (int m = 0; m < M; ++m) acc[m][0] = acc[m][1] = f32x2{0.0f, 0.0f};
        for (int k = 0; k < ROWS_PER_BLOCK; ++k) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b + (size_t)(k0 + k) * N + 4 * n4);
            const f32x2 lo{bv[0], bv[1]}, hi{bv[2], bv[3]};
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float am = as[k * M + m];
                acc[m][0] = __builtin_elementwise_fma(f32x2{am, am}, lo, acc[m][0]);
                acc[m][1] = __builtin_elementwise_fma(f32x2{am, am}, hi, acc[m][1]);
            }
        }
        for (int m = 0; m < M; ++m)
            *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * M + m) * N + 4 * n4) = f32x4{acc[m][0][0], acc[m][0][1], acc[m][1][0], acc[m][1][1]};
    } else {
        float acc[M][4];
        for (int m = 0; m < M; ++m) for (int j = 0; j < 4; ++j) acc[m][j] = 0.0f;
        for (int k = 0; k < ROWS_PER_BLOCK; ++k) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b + (size_t)(k0 + k) * N + 4 * n4);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float am = as[k * M + m];
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[m][j]) : "v"(am), "v"(bv[j]));     // (plain C: the SLP vectorizer packs these too)
            }
        }
        for (int m = 0; m < M; ++m)
            *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * M + m) * N + 4 * n4) = f32x4{acc[m][0], acc[m][1], acc[m][2], acc[m][3]};
    }
}

__global__ void join(const float* __restrict__ partial, float* __restrict__ out, int blocks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * N) return;
    float s = 0.0f;
    for (int b = 0; b < blocks; ++b) s += partial[(size_t)b * M * N + i];
    out[i] = s;
}

template <bool BF16>
__global__ void __launch_bounds__(256) partner(float* sink, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const float seed = (float)(threadIdx.x & 7) * 0.125f;
    if (BF16) {
        bf16x8 x, y;
        for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(seed + j); y[j] = (__bf16)(0.5f - 0.0625f * j); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i], 0, 0, 0);
    } else {
        const float x = seed, y = 0.5f;
        for (int it = 0; it < 4 * iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;           // (never: keeps the loop)
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 48;
    int n_cu = 256;
    CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    std::vector<float> ha((size_t)K * M), hb((size_t)K * N);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((float)(s >> 8) / 8388608.0f) - 1.0f; };
    for (float& v : ha) v = rnd();
    for (float& v : hb) v = rnd();
    const int blocks = K / ROWS_PER_BLOCK;
    float *a, *b, *partial, *out, *sink;
    CHECK(hipMalloc(&a, ha.size() * 4)); CHECK(hipMalloc(&b, hb.size() * 4));
    CHECK(hipMalloc(&partial, (size_t)blocks * M * N * 4)); CHECK(hipMalloc(&out, M * N * 4)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s_victim, s_partner;
    CHECK(hipStreamCreate(&s_victim)); CHECK(hipStreamCreate(&s_partner));
    std::vector<float> solo(M * N), got(M * N);
    int exit_code = 0;
    for (int packed = 1; packed >= 0; --packed) {
        auto run_victim = [&]() {
            if (packed) hipLaunchKernelGGL(victim<true>, dim3(blocks), dim3(128), 0, s_victim, a, b, partial);
            else hipLaunchKernelGGL(victim<false>, dim3(blocks), dim3(128), 0, s_victim, a, b, partial);
            hipLaunchKernelGGL(join, dim3((M * N + 255) / 256), dim3(256), 0, s_victim, partial, out, blocks);
        };
        run_victim();
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(solo.data(), out, M * N * 4, hipMemcpyDeviceToHost));
        run_victim();
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(got.data(), out, M * N * 4, hipMemcpyDeviceToHost));
        printf("%s victim alone, repeat identical: %s\n", packed ? "packed" : "scalar", memcmp(solo.data(), got.data(), M * N * 4) ? "NO" : "yes");
        for (int kind = 0; kind < 3; ++kind) {       // 0 nothing, 1 f32-input MFMA partner, 2 bf16 MFMA partner
            int bad_launches = 0, worst_count = 0, lane_lo = 1 << 30, lane_hi = -1;
            float worst = 0.0f;
            for (int l = 0; l < launches; ++l) {
                if (kind == 1) hipLaunchKernelGGL(partner<false>, dim3(n_cu), dim3(256), 0, s_partner, sink, 4000);
                if (kind == 2) hipLaunchKernelGGL(partner<true>, dim3(n_cu), dim3(256), 0, s_partner, sink, 4000);
                run_victim();
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(got.data(), out, M * N * 4, hipMemcpyDeviceToHost));
                int count = 0;
                for (int i = 0; i < M * N; ++i)
                    if (memcmp(&got[i], &solo[i], 4)) {
                        ++count;
                        const float d = got[i] > solo[i] ? got[i] - solo[i] : solo[i] - got[i];
                        if (d > worst) worst = d;
                        const int lane = ((i % N) / 4) & 63;
                        if (lane < lane_lo) lane_lo = lane;
                        if (lane > lane_hi) lane_hi = lane;
                    }
                if (count) { ++bad_launches; if (count > worst_count) worst_count = count; }
            }
            printf("  %s victim beside %-26s launches with differing values %2d / %d, values differing (max per launch) %4d of %d, largest difference %.3g",
                   packed ? "packed" : "scalar", kind == 0 ? "nothing" : kind == 1 ? "f32-input MFMA partner" : "bf16 MFMA partner",
                   bad_launches, launches, worst_count, M * N, worst);
            if (lane_hi >= 0) printf("  lanes %d..%d", lane_lo, lane_hi);
            printf("\n");
            if (bad_launches) exit_code = 1;
        }
    }
    return exit_code;
}
