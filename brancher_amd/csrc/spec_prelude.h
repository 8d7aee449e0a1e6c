// spec_prelude.h — first half of the hand-written frame around a program-specialised ELBO kernel.
//
// libbsvi turns a model program (include/bsvi.h: the memory-to-memory instruction stream the interpreter in
// elbo_kernel.hip walks) into straight-line HIP: every per-sample slot becomes a register, every instruction a
// few lines of arithmetic, the record loops are unrolled (specialize.cpp).  The generated translation unit is
//
//     #define SPEC_* ...            sizes of the program, launch bounds, variant
//     #include "spec_prelude.h"     this file: argument block, per-lane state, node arithmetic, noise
//     spec_body(...)                GENERATED: forward sweep, turn, reverse sweep of ONE Monte-Carlo sample
//     #include "spec_main.h"        the kernel: uniform-table prologue, spec_body, fixed-order reductions,
//                                   chain rule to theta, finalize, optimizer step, in-kernel training loop
//
// and is compiled for gfx950 with hiprtc when the program is first launched.  One lane = one Monte-Carlo sample,
// as in the interpreter, but nothing per-sample lives in LDS: no instruction fetch, no operand decode, no LDS
// round trip on the dependent path (brancher/variables.py:486-570,718-749 per node).
#pragma once
#include "bsvi_device.h"
#include "spec_args.h"

namespace bsvi {

// what the generated body reads of the argument block: the kernel copies these few words out of the kernarg segment
// right before the body, so that the other ~45 dwords of SpecArgs are not live (in scalar registers) across it
struct SpecBody {
    const float* noise;                  // this iteration's [n_noise][n_local], or null (diagnostic variant)
    float* samples_out;
    float* noise_out;
    float* fvalue_out;
    uint32_t n_local, seed_lo, seed_hi;
};

// per-lane state of the generated body
struct SpecLane {
    float f, lq;             // log p + entropy terms; log q (score term of the BlackBox estimator)
    uint32_t n, nc;          // local sample index; clamped to the shard (inactive lanes shadow the last sample)
    uint32_t nidx;           // global sample index: Philox counter
    uint32_t lane;
    uint32_t off_lo, off_hi; // Philox offset of this iteration
    uint32_t vz;             // always 0, but a per-lane value the compiler cannot prove uniform (SPEC_U)
    bool active;
};

__device__ __forceinline__ float spec_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float spec_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ---- Normal node with affine location, NormalVariable(loc = A*B + C, scale = S): BSVI_OP_NAFF -----------------
__device__ __forceinline__ float spec_naff_lp(float v, float loc, float S) {
    const float u = (v - loc) * spec_rcp(S);
    return -0.5f * (u * u) - spec_log(S) - kLogSqrt2Pi;
}
// a model log-probability term with constant weight w: value into f, adjoints of loc and S out
__device__ __forceinline__ void spec_naff_sink(float w, float v, float loc, float S, float& f, float& gloc, float& gS) {
    const float rS = spec_rcp(S);
    const float u = (v - loc) * rS;
    f += w * (-0.5f * (u * u) - spec_log(S) - kLogSqrt2Pi);
    gloc = w * u * rS;                   // d lp / d loc = (v - loc) / S^2
    gS = w * (u * u - 1.0f) * rS;
}
// reverse of the log-prob part of a posterior node: weight gw on log N(v | loc, S)
__device__ __forceinline__ void spec_naff_lp_bwd(float gw, float v, float loc, float S, float& gv, float& gloc, float& gS) {
    const float rS = spec_rcp(S);
    const float d = v - loc, t = d * (rS * rS);
    gv = -gw * t;
    gloc = gw * t;
    gS = gw * (d * t * rS - rS);
}

// ---- noise ---------------------------------------------------------------------------------------------------
// four standard normals of noise-row group g (rows 4g..4g+3): ONE Philox4x32-10 call, Box-Muller on v_sin/v_cos —
// the same stream as the interpreter's philox_normal (elbo_kernel.hip), so both engines draw the same samples
__device__ __forceinline__ void spec_normals4(const SpecBody& A, const SpecLane& T, uint32_t g,
                                              float& z0, float& z1, float& z2, float& z3) {
    const u32x4 x = philox4x32_10(T.nidx, g | 0x80000000u, T.off_lo, T.off_hi, A.seed_lo, A.seed_hi);
    box_muller_fast(x.x, x.y, z0, z1);
    box_muller_fast(x.z, x.w, z2, z3);
}
__device__ __forceinline__ PhiloxKey spec_key(const SpecBody& A, const SpecLane& T) {
    return PhiloxKey{T.nidx, A.seed_lo, A.seed_hi, T.off_lo, T.off_hi};
}

// ---- the uniform table and the per-lane gradient contributions -----------------------------------------------
// LDS image of a workgroup (floats):
//   [0, SPEC_U_PAD)                      uniform table U, then the observed data
//   WS  [W][SPEC_NUG_PAD]                per-wave sums of d f / d U, in completion order ("positions")
//   RED [SPEC_RED_FLOATS]                value / non-finite sums per wave, block totals, flags
//   TR  [W][SPEC_TE * 65]                per-wave transpose tile: the lanes of a wave store their contribution to
//                                        position p at TR[(p % TE) * 65 + lane]; every TE positions lane j adds up
//                                        row j (64 conflict-free reads, fixed order) — ~3 instructions per entry and
//                                        lane instead of a 10-instruction DPP reduction per entry
#ifndef SPEC_TE
#define SPEC_TE 64
#endif
#define SPEC_MAX_WAVES (SPEC_MAX_THREADS / 64)
#define SPEC_U_PAD ((SPEC_N_UNIFORM + SPEC_N_OBS + 3) / 4 * 4)
#define SPEC_NUG_PAD ((SPEC_N_UGRAD + SPEC_TE + 3) / 4 * 4)      /* flushes write whole tiles: room for the last one */
#define SPEC_RED_FLOATS (2 * SPEC_MAX_WAVES + 8)
#define SPEC_TR_FLOATS (SPEC_TE * 65)
#define SPEC_LDS_FLOATS (SPEC_U_PAD + SPEC_MAX_WAVES * SPEC_NUG_PAD + SPEC_RED_FLOATS + SPEC_MAX_WAVES * SPEC_TR_FLOATS)
__shared__ __attribute__((aligned(16))) float spec_lds[SPEC_LDS_FLOATS];

// A uniform-table read.  The address is lane-uniform, and left provably so the compiler moves every entry into a
// scalar register (ds_read + v_readfirstlane): ~90 of them on top of the argument block overflow the 102 SGPRs and
// spill through v_writelane / v_readlane (800 instructions of a 4 300-instruction kernel at BASELINE config 1).
// Adding the per-lane zero keeps the entries in vector registers; the reads still broadcast, and adjacent entries
// still merge into ds_read2 / ds_read_b128.
#define SPEC_U(k) spec_lds[(k) + T.vz]

// contribution of this lane to position `pos` (a literal)
#define SPEC_DU(pos, val) TRw[((pos) % SPEC_TE) * 65u + T.lane] = T.active ? (val) : 0.0f

// positions [base, base + count) are complete: lane j < count adds the 64 lane contributions of position base + j
__device__ __forceinline__ void spec_du_flush(float* TRw, float* WSw, uint32_t lane, uint32_t base, uint32_t count) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's tile stores have landed (LDS is in order)
    if (lane < count) {
        const float* row = TRw + lane * 65u;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
        for (uint32_t l = 0; l < 64u; l += 4u) { s0 += row[l]; s1 += row[l + 1u]; s2 += row[l + 2u]; s3 += row[l + 3u]; }
        WSw[base + lane] = (s0 + s1) + (s2 + s3);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next tile overwrites the rows
}

}  // namespace bsvi
