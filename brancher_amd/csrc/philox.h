// philox.h — counter-based RNG for in-kernel reparameterisation noise (device code).
//
// The reference draws noise through torch's global generator once per node per iteration
// (torch normal.py:83-86 `_standard_normal`, reached from brancher/distributions.py:122).
// Here every draw is a pure function of (seed, global sample index, noise row, iteration,
// attempt), so a Monte-Carlo shard on any GPU sees the same stream it would see on one GPU
// and the kernel never reads noise from HBM.  Philox4x32 (Salmon et al., SC'11) with SEVEN rounds since round 6: the paper's Table 2
// lists Philox4x32-7 as the fewest rounds that pass BigCrush ("Crush-resistant"; 10 is its default for a safety margin).  The draw is
// the bound of the dense path's two launches (a drawing wave per SIMD at ~140 vector instructions per four normals), and
// three rounds are ~20 of them: config 4 103.8 -> 101.1 us per iteration (profiles/r6/philox_rounds.txt).  Parity is always on the
// noise a kernel REPORTS (the oracle replays it), so nothing else moves; seeds of earlier rounds give other streams.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace bsvi {

struct u32x4 { uint32_t x, y, z, w; };

constexpr int kPhiloxRounds = 7;

__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < kPhiloxRounds; ++r) {
        // both halves of a product from ONE v_mad_u64_u32 (written as __umulhi + multiply the compiler emits a
        // quarter-rate v_mul_hi_u32 and a quarter-rate v_mul_lo_u32 per product: 40 instead of 20 per call)
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    return {c0, c1, c2, c3};
}

// (0,1) open interval, 24 random bits
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float r = sqrtf(-2.0f * logf(u01(a)));
    float s, c;
    sincosf(6.28318530717958647692f * u01(b), &s, &c);
    z0 = r * c;
    z1 = r * s;
}

}  // namespace bsvi
