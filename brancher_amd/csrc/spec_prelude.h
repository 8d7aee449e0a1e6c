// spec_prelude.h — first half of the hand-written frame around a program-specialised ELBO kernel.
//
// libbsvi turns a model program (include/bsvi.h: the memory-to-memory instruction stream the interpreter in
// elbo_kernel.hip walks) into straight-line HIP: every per-sample slot becomes a register, every instruction a
// few lines of arithmetic, the record loops are unrolled (specialize.cpp).  The generated translation unit is
//
//     #define SPEC_* ...            sizes of the program, launch bounds, variant
//     #include "spec_prelude.h"     this file: argument block, per-lane state, node arithmetic, noise, LDS image
//     spec_draw(...)                GENERATED: the standard normals of one Monte-Carlo sample (Philox, or the caller's)
//     spec_body(...)                GENERATED: forward sweep, turn, reverse sweep of that sample
//     #include "spec_main.h"        the kernel: uniform-table prologue, spec_draw + spec_body, fixed-order reductions,
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
    const float* f_weight;               // caller weights (diagnostic variant), or null
    const float* q_weight;
    uint32_t n_local, seed_lo, seed_hi;
};

// per-lane state of the generated body
struct SpecLane {
    float f, lq;             // log p + entropy terms; log q (score term of the BlackBox estimator)
    float gw;                // weight of this sample's grad f (1, or the caller's f_weight: diagnostic variant)
    uint32_t n, nc;          // local sample index; clamped to the shard (inactive lanes shadow the last sample)
    uint32_t nidx;           // global sample index: Philox counter
    uint32_t lane;
    uint32_t off_lo, off_hi; // Philox offset of this iteration
    uint32_t vz;             // always 0, but a per-lane value the compiler cannot prove uniform (SPEC_U)
    uint32_t tile;           // SPEC_TILE: float offset of this lane's column of the wave's transpose tile
    bool active;
};

// the standard normals of a sample, drawn BEFORE the barrier that publishes the uniform table (they do not depend on
// it): registers after inlining.  Programs with more noise rows than the generator keeps draw inside the body instead.
#ifndef SPEC_KEEP_NOISE
#define SPEC_KEEP_NOISE 0
#endif
struct SpecNoise { float z[SPEC_KEEP_NOISE ? SPEC_KEEP_NOISE : 1]; };

__device__ __forceinline__ float spec_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float spec_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ---- Normal node with affine location, NormalVariable(loc = A*B + C, scale = S): BSVI_OP_NAFF -----------------
// rS = 1/S and lS = log S come from the caller: for a lane-uniform scale (the usual case: a transformed parameter)
// they are entries of the uniform table's companions, computed once per iteration by the thread that owns the
// parameter; for a per-sample scale they are v_rcp_f32 / v_log_f32 of it.
__device__ __forceinline__ float spec_naff_lp(float v, float loc, float rS, float lS) {
    const float u = (v - loc) * rS;
    return -0.5f * (u * u) - lS - kLogSqrt2Pi;
}
// a model log-probability term with constant weight w: value into f, adjoints of loc and S out
__device__ __forceinline__ void spec_naff_sink(float w, float v, float loc, float rS, float lS, float& f, float& gloc, float& gS) {
    const float u = (v - loc) * rS;
    f += w * (-0.5f * (u * u) - lS - kLogSqrt2Pi);
    gloc = w * u * rS;                   // d lp / d loc = (v - loc) / S^2
    gS = w * (u * u - 1.0f) * rS;
}
// reverse of the log-prob part of a posterior node: weight gw on log N(v | loc, S)
__device__ __forceinline__ void spec_naff_lp_bwd(float gw, float v, float loc, float rS, float& gv, float& gloc, float& gS) {
    const float d = v - loc, t = d * (rS * rS);
    gv = -gw * t;
    gloc = gw * t;
    gS = gw * (d * t * rS - rS);
}

// ---- noise ---------------------------------------------------------------------------------------------------
// four standard normals of noise-row group g (rows 4g..4g+3): ONE Philox4x32 call (kPhiloxRounds rounds, philox.h), Box-Muller on v_sin/v_cos —
// the same stream as the interpreter's philox_normal (elbo_kernel.hip), so both engines draw the same samples
__device__ __forceinline__ void spec_normals4(const SpecBody& A, const SpecLane& T, uint32_t g,
                                              float& z0, float& z1, float& z2, float& z3) {
    const u32x4 x = philox4x32(T.nidx, g | 0x80000000u, T.off_lo, T.off_hi, A.seed_lo, A.seed_hi);
    box_muller_fast(x.x, x.y, z0, z1);
    box_muller_fast(x.z, x.w, z2, z3);
}
// a value the optimiser cannot see through
__device__ __forceinline__ uint32_t spec_opaque(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(x));
#endif
    return x;
}
__device__ __forceinline__ PhiloxKey spec_key(const SpecBody& A, const SpecLane& T) {
    return PhiloxKey{T.nidx, A.seed_lo, A.seed_hi, T.off_lo, T.off_hi};
}

// ---- LDS image of a workgroup (floats) -----------------------------------------------------------------------
//   U    [SPEC_U_PAD]                   uniform table U, then the observed data
//   UR   [SPEC_NU_PAD]                  1 / U[k]      } companions of the uniform table: what a Normal node needs
//   UL   [SPEC_NU_PAD]                  log U[k]      } of a lane-uniform scale
//   WS   [W][4 * SPEC_N_POS]            per-wave sums of d f / d U: position p ("completion order") holds the totals
//                                       of the wave's four 16-lane rows at WS[4p .. 4p+3]
//   RED  [SPEC_RED_FLOATS]              value / non-finite sums per wave, block totals, flags
//   PS   [5][SPEC_NP_PAD]               theta and the optimizer state [4][n_params]: the working copy of a launch
//   TAB                                 uniform entries (4 words each), CSR theta -> positions / uniform indices, masks
//   OWN  [min(NP, threads)][16]         what the thread that owns a parameter needs per iteration, packed for four
//                                       ds_read_b128: theta, optimizer state, its (<= 2) uniform entries
//   SCR                                 scratch of the grid reduction (several workgroups)
//   TR   [W][64 * 68]                   SPEC_TILE programs (at most 64 positions): per-wave transpose tile, see SPEC_DU
#ifndef SPEC_GENERIC_OWNERS
#define SPEC_GENERIC_OWNERS 1
#endif
#ifndef SPEC_ACCUMULATE_CHUNKS
#define SPEC_ACCUMULATE_CHUNKS 0
#endif
#define SPEC_MAX_WAVES (SPEC_MAX_THREADS / 64)
#define SPEC_U_PAD ((SPEC_N_UNIFORM + SPEC_N_OBS + 3) / 4 * 4)
#define SPEC_NU_PAD ((SPEC_N_UNIFORM + 3) / 4 * 4)
#ifndef SPEC_TILE
#define SPEC_TILE 0
#endif
#if SPEC_TILE
#define SPEC_WS_PAD ((SPEC_N_POS + 3) / 4 * 4 + 4)
#define SPEC_TR_STRIDE 68      /* 16-byte aligned rows; 16 lanes x 4 consecutive banks tile the 64 banks exactly */
#define SPEC_TR_FLOATS (64 * SPEC_TR_STRIDE)
#else
#define SPEC_WS_PAD (4 * SPEC_N_POS + 4)
#define SPEC_TR_FLOATS 0
#endif
#define SPEC_RED_FLOATS (2 * SPEC_MAX_WAVES + 8)
#define SPEC_NP_PAD ((SPEC_N_PARAMS + 3) / 4 * 4 + 4)
#define SPEC_TAB_WORDS ((4 * SPEC_N_UNIFORM + (2 * SPEC_N_PARAMS + 1) + 2 * SPEC_N_POS + 3) / 4 * 4 + 4)
#define SPEC_SCR_FLOATS (4 * (SPEC_N_POS + 2) + 4)
#define SPEC_OFF_UR SPEC_U_PAD
#define SPEC_OFF_UL (SPEC_OFF_UR + SPEC_NU_PAD)
#define SPEC_OFF_WS (SPEC_OFF_UL + SPEC_NU_PAD)
#define SPEC_OFF_RED (SPEC_OFF_WS + SPEC_MAX_WAVES * SPEC_WS_PAD)
#define SPEC_OFF_PS (SPEC_OFF_RED + SPEC_RED_FLOATS)
#define SPEC_OFF_TAB (SPEC_OFF_PS + 5 * SPEC_NP_PAD)
#define SPEC_OWN_ROWS (SPEC_N_PARAMS < SPEC_MAX_THREADS ? SPEC_N_PARAMS : SPEC_MAX_THREADS)
#define SPEC_OFF_OWN (SPEC_OFF_TAB + SPEC_TAB_WORDS)
#define SPEC_OWN_WORDS 20                  // an owner row: SpecOwn packed (spec_main.h)
#define SPEC_OFF_SCR (SPEC_OFF_OWN + SPEC_OWN_WORDS * SPEC_OWN_ROWS)
#define SPEC_OFF_TR (SPEC_OFF_SCR + SPEC_SCR_FLOATS)
#define SPEC_LDS_FLOATS (SPEC_OFF_TR + SPEC_MAX_WAVES * SPEC_TR_FLOATS)
__shared__ __attribute__((aligned(16))) float spec_lds[SPEC_LDS_FLOATS];

// A uniform-table read.  The address is lane-uniform, and left provably so the compiler moves every entry into a
// scalar register (ds_read + v_readfirstlane): ~90 of them on top of the argument block overflow the 102 SGPRs and
// spill through v_writelane / v_readlane.  Adding the per-lane zero keeps the entries in vector registers; the reads
// still broadcast, and adjacent entries still merge into ds_read2 / ds_read_b128.
#define SPEC_U(k) spec_lds[(k) + T.vz]
#define SPEC_UR(k) spec_lds[SPEC_OFF_UR + (k) + T.vz]
#define SPEC_UL(k) spec_lds[SPEC_OFF_UL + (k) + T.vz]

typedef float spec_f4 __attribute__((ext_vector_type(4)));

// Contribution of this lane to position `pos` (a literal).  Two schemes, chosen per program by the generator:
//
//  SPEC_TILE (programs with at most 64 positions — BASELINE configs 1 and 2): the lanes of a wave store their
//  contribution at TR[pos * 68 + lane] of the wave's transpose tile — one instruction — and ONE flush at the end of the
//  body has lane j add up row j with 16 conflict-free ds_read_b128, in a fixed order.
//
//  otherwise: four DPP adds leave every lane with the total of its 16-lane row (quad_perm, quad_perm, row_half_mirror,
//  row_mirror: a fixed order), and the row's lanes store it — same address, same value — to the row's cell of the
//  position, WS[4 pos + row].  Five instructions per contribution, but straight-line: with more than 64 positions a
//  tile needs flushes INSIDE the body, and those — 64 registers of reads between memory fences — cut the body into
//  regions the register allocator handles badly (2.5 x the registers at T = 60, spilled to scratch).
//
// A workgroup that walks several sample chunks (SPEC_ACCUMULATE_CHUNKS) accumulates into the cells instead of storing.
__device__ __forceinline__ float spec_row_sum(float x) {
    x += dpp_f<0xB1>(x);
    x += dpp_f<0x4E>(x);
    x += dpp_f<0x141>(x);
    x += dpp_f<0x140>(x);
    return x;
}
#if SPEC_TILE
#define SPEC_DU(pos, val) spec_lds[SPEC_OFF_TR + T.tile + (pos) * SPEC_TR_STRIDE] = T.active ? (val) : 0.0f
__device__ __forceinline__ void spec_du_flush(const float* TRw, float* WSw, uint32_t lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's tile stores have landed (LDS is in order)
    if (lane < SPEC_N_POS) {
        const spec_f4* row = reinterpret_cast<const spec_f4*>(TRw + lane * SPEC_TR_STRIDE);
        spec_f4 s = row[0];
#pragma unroll
        for (uint32_t q = 1; q < 16u; ++q) s += row[q];
        const float t = (s.x + s.y) + (s.z + s.w);
#if SPEC_ACCUMULATE_CHUNKS
        WSw[lane] += t;
#else
        WSw[lane] = t;
#endif
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next chunk overwrites the rows
}
#elif SPEC_ACCUMULATE_CHUNKS
#define SPEC_DU(pos, val) { float* const c_ = WSw + 4u * (pos) + (T.lane >> 4); *c_ = *c_ + spec_row_sum(T.active ? (val) : 0.0f); }
#else
#define SPEC_DU(pos, val) WSw[4u * (pos) + (T.lane >> 4)] = spec_row_sum(T.active ? (val) : 0.0f)
#endif
// floats of a wave's sums / cell of position `pos` in them
#if SPEC_TILE
#define SPEC_WS_CELLS SPEC_N_POS
#define SPEC_WS_CELL(pos) (pos)
#else
#define SPEC_WS_CELLS (4u * SPEC_N_POS)
#define SPEC_WS_CELL(pos) (4u * (pos))
#endif
// the total of position `pos` over the first `rows` waves' cells, waves in order (literal trip count and row offsets — the
// rows of all SPEC_MAX_WAVES waves exist, those past `rows` are read and not selected: one address, immediates, back to back)
__device__ __forceinline__ float spec_pos_total(const float* WS, uint32_t pos, uint32_t rows) {
    float s = 0.0f;
#pragma unroll
    for (uint32_t w = 0; w < SPEC_MAX_WAVES; ++w) {
#if SPEC_TILE
        const float t = WS[w * SPEC_WS_PAD + pos];
#else
        const spec_f4 q = *reinterpret_cast<const spec_f4*>(WS + w * SPEC_WS_PAD + 4u * pos);
        const float t = (q.x + q.y) + (q.z + q.w);
#endif
        s += w < rows ? t : 0.0f;
    }
    return s;
}

}  // namespace bsvi
