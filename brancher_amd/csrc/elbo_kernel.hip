// elbo_kernel.hip — fused ELBO forward+backward for gfx950 (MI355X), and the C ABI around it.
//
// One lane = one Monte-Carlo sample; a wavefront walks the compiled model program
// (include/bsvi.h) in lock-step, so control flow is wave-uniform and the program stream is
// fetched through the scalar cache.  Per workgroup:
//
//   prologue   U[k] = a + b*g(theta|const)  — every lane-uniform parameter transform, once
//   forward    q nodes: link -> SAMPLE (noise from HBM [row][N], or in-register Philox) -> ENTROPY;
//              p nodes (sinks): link -> LOGP and, in the same visit, their adjoints.  A sample's
//              values and adjoints live in its own LDS row (odd row length: the 64 lanes of a wave
//              hit 64 banks); f and log q stay in registers.
//   reverse    the posterior's sampling chain in reverse, hand-derived adjoints.  Storage modes:
//              lds+lane_acc — the row also carries one accumulator per parameter-sourced uniform
//                entry (no cross-lane traffic in the sweeps, one fixed-order lane reduction per launch)
//                and, when it fits, the node's noise; top-level Normal nodes run on handlers whose
//                addresses bsvi_program_create resolved beforehand (struct Aux);
//              lds+wave_sum — uniform adjoints are summed over the 64 lanes with DPP + v_readlane and
//                accumulated per wave; global — rows spill to a [slot][N] workspace.
//              No atomics anywhere: bitwise reproducible.
//   epilogue   per-workgroup partial sums -> workspace; `reduce_kernel` (one workgroup) adds
//              them in fixed order, applies dU/dtheta through a CSR map, and optionally fuses
//              the .mean()/sign (`finalize`) and the optimizer step.  `persistent_kernel` keeps
//              the whole optimisation loop inside one launch.
//
// This replaces, per iteration, the ~18 400 ATen dispatches the reference issues from
// brancher/variables.py:486-570,718-749,843-870 + gradient_estimators.py:29-44 +
// loss.backward() (inference.py:100) + optimizers.py:69-70.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <type_traits>
#include <string>
#include <vector>

#include "bsvi.h"
#include "bsvi_device.h"
#include "bsvi_internal.h"

namespace bsvi {


struct KParams {
    const uint4* code;
    const uint4* aux;       // [n_code][4]: pre-resolved addresses of the instructions marked kFastFlag (library-internal)
    const bsvi_uniform_entry* uniform;
    const float* consts;
    const float* params;
    const float* obs;
    const float* noise;
    float* samples_out;
    float* noise_out;
    float* fvalue_out;
    // caller-supplied per-sample weights of the gradient (bsvi_elbo_args::f_weight_dev / q_weight_dev): sum_n a_n grad f_n +
    // b_n grad log q_n instead of the estimator's own a_n = 1, b_n = stopgrad(f_n); NULL otherwise
    const float* f_weight;
    const float* q_weight;
    float* partials;   // [grid][2 + n_uniform_grad]
    float* zglobal;    // ZG variant: [2 * n_slots][n_pad]
    unsigned long long* stamps;   // diagnostic build only: phase time stamps of block 0 (or NULL)
    // single-workgroup launches of bsvi_elbo_fwd_bwd: the chain rule to theta runs in the kernel's own epilogue
    // and the output block is written here (no reduce_kernel launch); NULL otherwise
    float* fuse_out;
    const uint32_t* pu_ptr;
    const uint32_t* pu_idx;
    uint32_t n_params;
    uint32_t n_code, n_uniform, n_uniform_grad, n_slots, n_noise, n_obs, estimator;
    uint32_t n_local, n_global, sample_base, n_pad;
    uint32_t lpw;           // lanes of a wave that carry samples (64; fewer when a full wave's rows do not fit LDS)
    uint32_t stash;         // lds+lane_acc: rows carry one cell per noise row; the reverse sweep reads eps back
                            // instead of regenerating it (Philox + Box-Muller are ~8 % of a wave's instructions)
    uint32_t seed_lo, seed_hi, offset_lo, offset_hi;
    const unsigned long long* offset_dev;   // added to the offset when non-null (graph-replayed multi-GPU step)
    // shares of the program over workgroups (multi-workgroup launches of elbo_kernel): workgroup b runs share
    // b % n_shares on sample group b / n_shares (bsvi_program_set_shares); 0 / 1 = the whole program
    uint32_t n_shares;
    uint32_t share_n_code[8];
    const uint4* share_code[8];
    const uint4* share_aux[8];
};

// ---------------------------------------------------------------------------------------
// dynamic LDS of every kernel in this file.  Declared at file scope and always indexed
// directly, so that every access is a real ds_read/ds_write (a pointer that travels through
// a struct degrades to flat loads, which cost a full memory round trip per operand).
//   [0, n_uniform)                      U      lane-uniform table (params/consts, transformed)
//   [n_uniform, +n_obs)                 OBS    observed data, staged once per launch
//   uadj  [n_uniform_grad][n_waves]            per-wave partial sums of dU
//   red   [4 * n_waves]
//   rows  [nthreads][2 * n_slots + 1]          one row per sample: (value, adjoint) pairs of its
//                                              slots.  The odd row length makes the 64 lanes
//                                              of a wave hit distinct banks for any slot, and
//                                              makes a slot's address lane_row + 8 * slot —
//                                              independent of the launch geometry, so operands
//                                              are resolved to byte offsets at lowering time.
//                                              (ZG: slots live in a global [slot][n_pad] array)
// ---------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) float g_lds[];

struct Lay {   // wave-uniform layout
    uint32_t obs, uadj, red, rows;      // float offsets
    uint32_t aux;                       // SM_LACC: float offset of the LDS copy of the Aux table (behind the rows)
    uint32_t row_words;                 // 2 * n_slots + 1 (+ lane accumulators), odd
    uint32_t uacc, dummy;               // SM_LACC: byte offsets inside a row of the per-lane
                                        // dU accumulators and of the write-only dummy cell
    uint32_t ugrad_bytes;               // uniform entries below this byte offset carry gradients
    uint32_t nthreads, n_waves;
    uint32_t lpw, rpw;                  // sample lanes per wave; rows per wave (lpw, +1 shared dummy row if lpw < 64)
};

// per-lane state — lives in registers: never take its address across a call
struct Lane {
    float f, lq, fweight, mask;      // fweight: the score term's weight of a log q adjoint ALREADY times the lane's mask
    uint32_t n, nc, tid, wave, lane;
    uint32_t zrow;              // byte address of this sample's slot row in LDS
    uint32_t nidx;              // global sample index (Philox counter)
    uint32_t cached_group;      // wave-uniform
    float c0, c1, c2, c3;       // cached normals of that group
};

// standard normal for noise row `row`: rows 4g..4g+3 share one Philox call (the forward sweep
// walks the rows upwards, the reverse sweep downwards: either way 1 call per 4 draws)
__device__ __forceinline__ float philox_normal(const KParams& K, Lane& T, uint32_t row) {
    const uint32_t group = row >> 2;
    if (group != T.cached_group) {
        const u32x4 x = philox4x32(T.nidx, group | 0x80000000u, K.offset_lo, K.offset_hi, K.seed_lo, K.seed_hi);
        float z0, z1, z2, z3;
        box_muller_fast(x.x, x.y, z0, z1);
        box_muller_fast(x.z, x.w, z2, z3);
        T.c0 = z0; T.c1 = z1; T.c2 = z2; T.c3 = z3;
        T.cached_group = group;
    }
    // select on VALUES: a conditional over the struct fields themselves is an lvalue (a select of
    // addresses), which pins the four cached normals in scratch memory — one flat load per draw
    const uint32_t j = row & 3u;
    const float c0 = T.c0, c1 = T.c1, c2 = T.c2, c3 = T.c3;
    const float lo = (j & 1u) ? c1 : c0, hi = (j & 1u) ? c3 : c2;
    return (j & 2u) ? hi : lo;
}

// Program tables are immutable for the lifetime of a launch and are addressed with
// wave-uniform indices: read them through the constant address space so that hipcc emits
// scalar loads (one s_load_dwordx8 per instruction through the scalar cache).
#if defined(__HIP_DEVICE_COMPILE__)
#define BSVI_CONST_AS __attribute__((address_space(4)))
#else
#define BSVI_CONST_AS   /* host pass: never executed */
#endif
struct Insn { uint32_t w0, dst, a, b, c, s, imm0, imm1; };
__device__ __forceinline__ Insn ld_insn(const uint4* code, uint32_t pc) {
    return *((const BSVI_CONST_AS Insn*)(code) + pc);
}
__device__ __forceinline__ bsvi_record ld_record(const bsvi_record* recs, uint32_t i) {
    return *((const BSVI_CONST_AS bsvi_record*)(recs) + i);
}

// Library-internal companion of an instruction (80 bytes), built by bsvi_program_create: everything the
// interpreter would otherwise derive from the operand words with scalar ALU work on every visit, for
// the top-level NAFF instructions whose adjoint cells are pairwise distinct (marked kFastFlag).
// Operand order: dst, a, b, c, s.
//   w0, imm0, imm1   copies of the instruction's words (w0 of EVERY instruction, marked or not): the sweeps
//            walk this table and read the 32-byte instruction on the generic path only
//   row      noise row of a sampled destination
//   off[k]   byte offset of the value (uniform region, or inside the lane's row)
//   mask[k]  ~0 for a per-lane operand, 0 for a lane-uniform one: address = off + (lane_row & mask)
//   cell[k]  byte offset INSIDE the lane's row of the operand's adjoint cell in the lds+lane_acc layout
//            (slot adjoint, per-lane dU accumulator, or the write-only dummy)
// The table is copied to LDS once per launch and an entry is fetched one visit ahead with five
// ds_read_b128 from a wave-uniform address, into VECTOR registers, on purpose: (1) the kernels are
// short of scalar registers, not of vector ones — 40 more SGPRs for a double-buffered entry spill around
// every visit; (2) scalar loads share the lgkm counter with LDS and return out of order, so a scalar
// prefetch in flight forces every LDS wait of the visit to drain it; (3) global loads of the entry
// (tried) cost an L1/L2 round trip longer than a visit.  The single-SIMD wave stream is issue-bound
// (every instruction, scalar or vector, costs ~4.7 cycles), so this table roughly halves the cost of a node.
struct Aux { uint32_t w0, imm0, imm1, row, off[5], mask[5], cell[5], eps_cell; };   // eps_cell: stash of the node's noise
constexpr uint32_t kAuxWords = 20;
constexpr uint32_t kFastFlag = 4u;      // rflags bit set in the DEVICE copy of the code (not part of the ABI)
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
struct AuxRaw { u32x4v q0, q1, q2, q3, q4; };      // an entry in flight: five 16-byte LDS reads
// Issue the five reads of entry `pc` (wave-uniform address: broadcast reads, no bank conflicts; 16-byte aligned:
// L.aux is a multiple of 4 floats and an entry is 80 bytes).  Hand-written: left to the compiler some call
// sites are scalarised and re-merged into nine 4-byte-aligned ds_read2_b32.  The compiler's own lgkmcnt
// bookkeeping stays conservative-correct (LDS returns in order; extra older operations only make its counted
// waits stricter); the consumer calls aux_wait() before touching the entry.
__device__ __forceinline__ AuxRaw ld_aux(uint32_t aux_float_offset, uint32_t pc) {
    AuxRaw R;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t addr = (aux_float_offset + pc * kAuxWords) * 4u;
    asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\t"
                 "ds_read_b128 %3, %5 offset:48\n\tds_read_b128 %4, %5 offset:64"
                 : "=&v"(R.q0), "=&v"(R.q1), "=&v"(R.q2), "=&v"(R.q3), "=&v"(R.q4) : "v"(addr) : "memory");
#else
    const u32x4v* p = reinterpret_cast<const u32x4v*>(&g_lds[aux_float_offset + pc * kAuxWords]);
    R.q0 = p[0]; R.q1 = p[1]; R.q2 = p[2]; R.q3 = p[3]; R.q4 = p[4];
#endif
    return R;
}
// wait for the entry's reads (issued a whole visit earlier) and unpack it
__device__ __forceinline__ Aux aux_wait(AuxRaw R) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R.q0), "+v"(R.q1), "+v"(R.q2), "+v"(R.q3), "+v"(R.q4));
#endif
    Aux X;
    X.w0 = R.q0.x; X.imm0 = R.q0.y; X.imm1 = R.q0.z; X.row = R.q0.w;
    X.off[0] = R.q1.x; X.off[1] = R.q1.y; X.off[2] = R.q1.z; X.off[3] = R.q1.w;
    X.off[4] = R.q2.x; X.mask[0] = R.q2.y; X.mask[1] = R.q2.z; X.mask[2] = R.q2.w;
    X.mask[3] = R.q3.x; X.mask[4] = R.q3.y; X.cell[0] = R.q3.z; X.cell[1] = R.q3.w;
    X.cell[2] = R.q4.x; X.cell[3] = R.q4.y; X.cell[4] = R.q4.z; X.eps_cell = R.q4.w;
    return X;
}
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// log of a normal positive float (a scale): v_log_f32 * ln 2.  (`__logf` expands to a 12-instruction
// sequence with denormal scaling and a compensated product)
__device__ __forceinline__ float log_pos(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

enum { SM_WSUM = 0, SM_LACC = 1, SM_ZG = 2 };
#define ZG (SM == SM_ZG)
// ---- operand access.  An operand word is  byte_offset | walks<<30 | per_lane<<31  (resolved by
// the lowering): uniform entries / observed data are read at byte_offset of the uniform region
// (LDS offset 0), slots at lane_row + byte_offset, the adjoint of a slot 4 bytes further.
// ONE ds_read per operand, no branches. ---------------------------------------------------------
// Byte addresses are ABSOLUTE LDS addresses: the kernels that run elbo_block declare no static LDS, so the
// dynamic array g_lds starts at 0 (checked once per launch in elbo_block).  Going through `g_lds + offset`
// instead costs one `v_add_u32 v, 0, v` per access: the base is only known to be 0 after instruction selection.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) float lds_float_t;
__device__ __forceinline__ float lds_ld(uint32_t byte_addr) { return *reinterpret_cast<lds_float_t*>(byte_addr); }
__device__ __forceinline__ void lds_st(uint32_t byte_addr, float v) { *reinterpret_cast<lds_float_t*>(byte_addr) = v; }
__device__ __forceinline__ uint32_t lds_base_of_dynamic_array() {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)g_lds;
}
#else
__device__ __forceinline__ float lds_ld(uint32_t byte_addr) { return g_lds[byte_addr >> 2]; }
__device__ __forceinline__ void lds_st(uint32_t byte_addr, float v) { g_lds[byte_addr >> 2] = v; }
__device__ __forceinline__ uint32_t lds_base_of_dynamic_array() { return 0; }
#endif
__device__ __forceinline__ uint32_t opnd_offset(uint32_t o, uint32_t e) {
    const uint32_t per_lane = o >> 31;
    const uint32_t step = ((o >> 30) & 1u) * (4u + 4u * per_lane);      // 8 bytes per slot, 4 per entry
    return (o & 0x3FFFFFFFu) + e * step;
}
template <int SM>
__device__ __forceinline__ float ld_opnd(const KParams& K, const Lane& T, uint32_t o, uint32_t e) {
    const uint32_t off = opnd_offset(o, e);
    const uint32_t mask = (uint32_t)((int32_t)o >> 31);
    if (ZG) {
        if (mask) return K.zglobal[(size_t)(off >> 3) * K.n_pad + T.n];
        return lds_ld(off);
    }
    return lds_ld(off + (T.zrow & mask));
}
// slots addressed by a destination / slot operand
template <int SM>
__device__ __forceinline__ float ld_slot(const KParams& K, const Lane& T, uint32_t off) {
    return ZG ? K.zglobal[(size_t)(off >> 3) * K.n_pad + T.n] : lds_ld(T.zrow + off);
}
template <int SM>
__device__ __forceinline__ void st_slot(const KParams& K, const Lane& T, uint32_t off, float v) {
    if (ZG) K.zglobal[(size_t)(off >> 3) * K.n_pad + T.n] = v;
    else lds_st(T.zrow + off, v);
}
template <int SM>
__device__ __forceinline__ float ld_adj(const KParams& K, const Lane& T, uint32_t off) {
    return ZG ? K.zglobal[((size_t)K.n_slots + (off >> 3)) * K.n_pad + T.n] : lds_ld(T.zrow + off + 4u);
}
template <int SM>
__device__ __forceinline__ void st_adj(const KParams& K, const Lane& T, uint32_t off, float v) {
    if (ZG) K.zglobal[((size_t)K.n_slots + (off >> 3)) * K.n_pad + T.n] = v;
    else lds_st(T.zrow + off + 4u, v);
}
// scatter an adjoint to an operand: slots accumulate per lane; lane-uniform values are summed
// over the wave first (DPP) and accumulated per wave — no atomics, fixed order
template <int SM>
__device__ __forceinline__ void add_adj(const KParams& K, const Lay& L, const Lane& T, uint32_t o, uint32_t e, float g) {
    const uint32_t off = opnd_offset(o, e);
    if (SM == SM_LACC) {
        // every operand has a per-lane adjoint cell: its slot's, its uniform entry's accumulator,
        // or (constants, observed data) a write-only dummy — one branch-free read-modify-write,
        // no cross-lane traffic; the accumulators are summed over lanes once per launch
        const uint32_t cell = (o >> 31) ? off + 4u : ((off < L.ugrad_bytes) ? L.uacc + off : L.dummy);
        lds_st(T.zrow + cell, lds_ld(T.zrow + cell) + g);
        return;
    }
    if (o >> 31) {
        st_adj<SM>(K, T, off, ld_adj<SM>(K, T, off) + g);
    } else if (off < L.ugrad_bytes) {
        const float tot = wave_sum(g);
        if (T.lane == 0) g_lds[L.uadj + (off >> 2) * L.n_waves + T.wave] += tot;
    }
}

// SM_LACC with BSVI_R_NOALIAS: the five adjoint cells of a NAFF instruction are distinct, so the
// read-modify-writes can be issued as 5 reads, ONE wait, 5 writes instead of 5 dependent round trips
__device__ __forceinline__ uint32_t adj_cell(const Lay& L, uint32_t o, uint32_t e) {
    const uint32_t off = opnd_offset(o, e);
    return (o >> 31) ? off + 4u : ((off < L.ugrad_bytes) ? L.uacc + off : L.dummy);
}
__device__ __forceinline__ void add_adj5(const Lay& L, const Lane& T, uint32_t od_, uint32_t oa_, uint32_t ob_, uint32_t oc_,
                                         uint32_t os_, uint32_t e, float gd, float ga, float gb, float gc, float gs) {
    const uint32_t cd = T.zrow + adj_cell(L, od_, e), ca = T.zrow + adj_cell(L, oa_, e);
    const uint32_t cb = T.zrow + adj_cell(L, ob_, e), cc = T.zrow + adj_cell(L, oc_, e);
    const uint32_t cs = T.zrow + adj_cell(L, os_, e);
    const float od = lds_ld(cd), oa = lds_ld(ca), ob = lds_ld(cb), oc = lds_ld(cc), os = lds_ld(cs);
    lds_st(cd, od + gd);
    lds_st(ca, oa + ga);
    lds_st(cb, ob + gb);
    lds_st(cc, oc + gc);
    lds_st(cs, os + gs);
}

__device__ __forceinline__ PhiloxKey philox_key(const KParams& K, const Lane& T) {
    return PhiloxKey{T.nidx, K.seed_lo, K.seed_hi, K.offset_lo, K.offset_hi};
}

// ---- forward of one instruction at element e -----------------------------------------------
// NODES=false: only the arithmetic instructions (re-materialising temps for the reverse sweep)
// OUT: the launch wants samples / noise written out (diagnostic and API-edge launches only)
// GEN: the program contains instructions other than NAFF (launch-time property of the program);
// with GEN=false every generic path — Beta/Binomial/... node math, unary functions — is compiled
// out, which keeps the Normal-only interpreter small enough to live in registers
template <int SM, bool GEN>
__device__ __forceinline__ void exec_backward(const KParams& K, const Lay& L, Lane& T, const Insn& I, uint32_t e);

// A model log-probability term N(value | A*B + C, S) with a constant weight, finished in one visit:
// operands are loaded once, the value goes to f and the adjoints straight to their operands.
template <int SM>
__device__ __forceinline__ void naff_sink(const KParams& K, const Lay& L, Lane& T, const Insn& I, uint32_t e) {
    const float A = ld_opnd<SM>(K, T, I.a, e), B = ld_opnd<SM>(K, T, I.b, e);
    const float Cc = ld_opnd<SM>(K, T, I.c, e), S = ld_opnd<SM>(K, T, I.s, e);
    const float v = ld_opnd<SM>(K, T, I.dst, e);
    const float w = __uint_as_float(I.imm0);
    const float rS = __builtin_amdgcn_rcpf(S), logS = __logf(S);
    const float loc = A * B + Cc;
    const float u = (v - loc) * rS;
    T.f += w * (-0.5f * (u * u) - logS - kLogSqrt2Pi);
    const float gw = w * T.mask;
    const float gloc = gw * u * rS;                 // d lp / d loc = (v - loc) / S^2
    const float gS = gw * (u * u - 1.0f) * rS;
    if (SM == SM_LACC && ((I.w0 >> 24) & BSVI_R_NOALIAS)) {
        add_adj5(L, T, I.dst, I.a, I.b, I.c, I.s, e, -gloc, gloc * B, gloc * A, gloc, gS);
    } else {
        add_adj<SM>(K, L, T, I.dst, e, -gloc);
        add_adj<SM>(K, L, T, I.a, e, gloc * B);
        add_adj<SM>(K, L, T, I.b, e, gloc * A);
        add_adj<SM>(K, L, T, I.c, e, gloc);
        add_adj<SM>(K, L, T, I.s, e, gS);
    }
}

// ---- pre-resolved NAFF handlers (lds+lane_acc storage, no diagnostic outputs) ----------------
// Same arithmetic as naff_sink / exec_forward / exec_backward below, addresses from the Aux entry
// (which the sweep prefetched one visit ahead).
struct FastAddr { uint32_t v[5], c[5]; };
__device__ __forceinline__ FastAddr fast_addr(const Lane& T, const Aux& X) {
    FastAddr F;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        F.v[k] = X.off[k] + (T.zrow & X.mask[k]);
        F.c[k] = T.zrow + X.cell[k];
    }
    return F;
}
__device__ __forceinline__ void fast_sink(Lane& T, const Aux& X) {
    const FastAddr F = fast_addr(T, X);
    const float w = __uint_as_float(X.imm0);
    const float v = lds_ld(F.v[0]), A = lds_ld(F.v[1]), B = lds_ld(F.v[2]), Cc = lds_ld(F.v[3]), S = lds_ld(F.v[4]);
    const float od = lds_ld(F.c[0]), oa = lds_ld(F.c[1]), ob = lds_ld(F.c[2]), oc = lds_ld(F.c[3]), os = lds_ld(F.c[4]);
    const float rS = __builtin_amdgcn_rcpf(S), logS = log_pos(S);
    const float loc = A * B + Cc;
    const float u = (v - loc) * rS;
    T.f += w * (-0.5f * (u * u) - logS - kLogSqrt2Pi);
    const float gw = w * T.mask;
    const float gloc = gw * u * rS;                 // d lp / d loc = (v - loc) / S^2
    const float gS = gw * (u * u - 1.0f) * rS;
    lds_st(F.c[0], od - gloc);
    lds_st(F.c[1], oa + gloc * B);
    lds_st(F.c[2], ob + gloc * A);
    lds_st(F.c[3], oc + gloc);
    lds_st(F.c[4], os + gS);
}
__device__ __forceinline__ void fast_forward(const KParams& K, Lane& T, const Aux& X, uint32_t w0) {
    const uint32_t flags = (w0 >> 8) & 0xFFu;
    const FastAddr F = fast_addr(T, X);
    const float w_lp = __uint_as_float(X.imm0), w_ent = __uint_as_float(X.imm1);
    const float A = lds_ld(F.v[1]), B = lds_ld(F.v[2]), Cc = lds_ld(F.v[3]), S = lds_ld(F.v[4]);
    const float loc = A * B + Cc;
    float v;
    if (flags & BSVI_F_SAMPLE) {
        const uint32_t row = uniform_u32(X.row);
        const float eps = K.noise ? K.noise[(size_t)row * K.n_local + T.nc] : philox_normal(K, T, row);
        v = loc + eps * S;
        lds_st(F.v[0], v);
        if (K.stash) lds_st(T.zrow + X.eps_cell, eps);
    } else {
        v = lds_ld(F.v[0]);
    }
    const float logS = log_pos(S);
    if (flags & BSVI_F_ENT) T.f += w_ent * (kHalfLog2PiE + logS);
    if (flags & (BSVI_F_LOGP | BSVI_F_WF)) {
        const float u = (v - loc) * __builtin_amdgcn_rcpf(S);
        const float lp = -0.5f * (u * u) - logS - kLogSqrt2Pi;
        T.f += w_lp * lp;
        if (flags & BSVI_F_WF) T.lq += lp;          // score term of the BlackBox estimator (gradient_estimators.py:33)
    }
}
__device__ __forceinline__ void fast_backward(const KParams& K, Lane& T, const Aux& X, uint32_t w0) {
    const uint32_t flags = (w0 >> 8) & 0xFFu;
    const FastAddr F = fast_addr(T, X);
    const float w_lp = __uint_as_float(X.imm0), w_ent = __uint_as_float(X.imm1);
    const float v = lds_ld(F.v[0]), A = lds_ld(F.v[1]), B = lds_ld(F.v[2]), Cc = lds_ld(F.v[3]), S = lds_ld(F.v[4]);
    const float od = lds_ld(F.c[0]), oa = lds_ld(F.c[1]), ob = lds_ld(F.c[2]), oc = lds_ld(F.c[3]), os = lds_ld(F.c[4]);
    const float loc = A * B + Cc;
    const float rS = __builtin_amdgcn_rcpf(S);
    float gloc = 0.0f, gS = 0.0f, gv = 0.0f;
    if (flags & (BSVI_F_LOGP | BSVI_F_WF)) {
        const float gw = w_lp * T.mask + ((flags & BSVI_F_WF) ? T.fweight : 0.0f);
        const float d = v - loc, t = d * (rS * rS);
        gv = -gw * t;
        gloc = gw * t;
        gS = gw * (d * t * rS - rS);
    }
    if (flags & BSVI_F_ENT) gS += w_ent * T.mask * rS;
    float gdst = gv;
    if (flags & BSVI_F_SAMPLE) {
        // a sampled latent's own cell holds its incoming adjoint: read, folded into loc/scale, written back unchanged
        const float zb = od + gv;
        float eps;
        if (K.stash) {
            eps = lds_ld(T.zrow + X.eps_cell);
        } else {
            const uint32_t row = uniform_u32(X.row);
            eps = K.noise ? K.noise[(size_t)row * K.n_local + T.nc] : philox_normal(K, T, row);
        }
        gloc += zb;
        gS += zb * eps;
        gdst = 0.0f;
    }
    lds_st(F.c[0], od + gdst);
    lds_st(F.c[1], oa + gloc * B);
    lds_st(F.c[2], ob + gloc * A);
    lds_st(F.c[3], oc + gloc);
    lds_st(F.c[4], os + gS);
}

template <int SM, bool OUT, bool NODES, bool GEN>
__device__ __forceinline__ void exec_forward(const KParams& K, const Lay& L, Lane& T, const Insn& I, uint32_t e) {
    const uint32_t op = I.w0 & 0xFFu, flags = (I.w0 >> 8) & 0xFFu;
    if (op == BSVI_OP_NAFF) {
        if (!NODES) return;
        const float A = ld_opnd<SM>(K, T, I.a, e), B = ld_opnd<SM>(K, T, I.b, e);
        const float Cc = ld_opnd<SM>(K, T, I.c, e), S = ld_opnd<SM>(K, T, I.s, e);
        const float loc = A * B + Cc;
        float v;
        if (flags & BSVI_F_SAMPLE) {
            const uint32_t doff = opnd_offset(I.dst, e), row = doff >> 3;
            const float eps = K.noise ? K.noise[(size_t)row * K.n_local + T.nc] : philox_normal(K, T, row);
            v = (flags & BSVI_F_GIVEN) ? eps : loc + eps * S;
            st_slot<SM>(K, T, doff, v);
            if (OUT && T.mask != 0.0f) {
                if (K.samples_out) K.samples_out[(size_t)row * K.n_local + T.n] = v;
                if (K.noise_out) K.noise_out[(size_t)row * K.n_local + T.n] = eps;
            }
        } else {
            v = ld_opnd<SM>(K, T, I.dst, e);
        }
        // hardware log / reciprocal (1 ulp): the per-sample chain is issue-bound, not throughput-bound
        const float logS = __logf(S);
        if (flags & BSVI_F_ENT) T.f += __uint_as_float(I.imm1) * (kHalfLog2PiE + logS);
        if (flags & (BSVI_F_LOGP | BSVI_F_WF)) {
            const float u = (v - loc) * __builtin_amdgcn_rcpf(S);
            const float lp = -0.5f * (u * u) - logS - kLogSqrt2Pi;
            T.f += __uint_as_float(I.imm0) * lp;
            if (flags & BSVI_F_WF) T.lq += lp;
        }
    } else if (op == BSVI_OP_BIN) {
        const float a = ld_opnd<SM>(K, T, I.a, e), b = ld_opnd<SM>(K, T, I.b, e);
        float y;
        switch (flags) {
        case BSVI_B_ADD: y = a + b; break;
        case BSVI_B_SUB: y = a - b; break;
        case BSVI_B_MUL: y = a * b; break;
        case BSVI_B_DIV: y = a / b; break;
        case BSVI_B_POW: y = GEN ? pow_ff(a, b) : a; break;
        default: y = (a == b) ? 1.0f : 0.0f; break;
        }
        st_slot<SM>(K, T, opnd_offset(I.dst, e), y);
    } else if (op == BSVI_OP_UN) {
        const float a = ld_opnd<SM>(K, T, I.a, e);
        st_slot<SM>(K, T, opnd_offset(I.dst, e), unop<GEN>(flags, a, __uint_as_float(I.imm0)));
    } else if (GEN && op == BSVI_OP_NODE) {
        if (!NODES) return;
        const int dist = (int)((I.w0 >> 16) & 0xFFu);
        const float p0 = ld_opnd<SM>(K, T, I.a, e), p1 = ld_opnd<SM>(K, T, I.b, e);
        float v;
        if (flags & BSVI_F_SAMPLE) {
            const uint32_t doff = opnd_offset(I.dst, e), row = doff >> 3;
            float noise;
            if (K.noise) {
                noise = K.noise[(size_t)row * K.n_local + T.nc];
                v = (flags & BSVI_F_GIVEN) ? noise : sample_from_noise_generic(dist, p0, p1, noise);
            } else {
                const float2 d = philox_draw(philox_key(K, T), dist, p0, p1, row);
                v = d.x;
                noise = d.y;
            }
            st_slot<SM>(K, T, doff, v);
            if (OUT && T.mask != 0.0f) {
                if (K.samples_out) K.samples_out[(size_t)row * K.n_local + T.n] = v;
                if (K.noise_out) K.noise_out[(size_t)row * K.n_local + T.n] = noise;
            }
        } else {
            v = ld_opnd<SM>(K, T, I.dst, e);
        }
        if (flags & BSVI_F_ENT) T.f += __uint_as_float(I.imm1) * entropy_generic(dist, p0, p1);
        if (flags & (BSVI_F_LOGP | BSVI_F_WF)) {
            const float lp = logp_generic(dist, v, p0, p1);
            T.f += __uint_as_float(I.imm0) * lp;
            if (flags & BSVI_F_WF) T.lq += lp;
        }
    }
}

// ---- reverse of one instruction at element e -------------------------------------------------
template <int SM, bool GEN>
__device__ __forceinline__ void exec_backward(const KParams& K, const Lay& L, Lane& T, const Insn& I, uint32_t e) {
    const uint32_t op = I.w0 & 0xFFu, flags = (I.w0 >> 8) & 0xFFu;
    if (op == BSVI_OP_NAFF) {
        const float A = ld_opnd<SM>(K, T, I.a, e), B = ld_opnd<SM>(K, T, I.b, e);
        const float Cc = ld_opnd<SM>(K, T, I.c, e), S = ld_opnd<SM>(K, T, I.s, e);
        const float v = ld_opnd<SM>(K, T, I.dst, e);     // SAMPLE: DST is the latent's own slot
        const float loc = A * B + Cc;
        const float rS = __builtin_amdgcn_rcpf(S);
        float gloc = 0.0f, gS = 0.0f, gv = 0.0f;
        if (flags & (BSVI_F_LOGP | BSVI_F_WF)) {
            const float gw = __uint_as_float(I.imm0) * T.mask + ((flags & BSVI_F_WF) ? T.fweight : 0.0f);
            const float d = v - loc, t = d * (rS * rS);
            gv = -gw * t;
            gloc = gw * t;
            gS = gw * (d * t * rS - rS);
        }
        if (flags & BSVI_F_ENT) gS += __uint_as_float(I.imm1) * T.mask * rS;
        const bool sampled = (flags & BSVI_F_SAMPLE) != 0;
        if (SM == SM_LACC && ((I.w0 >> 24) & BSVI_R_NOALIAS)) {
            // batched: the five adjoint cells are read together (one wait); a sampled latent's own
            // cell is read for its incoming adjoint and written back unchanged
            const uint32_t cd = T.zrow + adj_cell(L, I.dst, e), ca = T.zrow + adj_cell(L, I.a, e);
            const uint32_t cb = T.zrow + adj_cell(L, I.b, e), cc = T.zrow + adj_cell(L, I.c, e);
            const uint32_t cs = T.zrow + adj_cell(L, I.s, e);
            const float od = lds_ld(cd), oa = lds_ld(ca), ob = lds_ld(cb), oc = lds_ld(cc), os = lds_ld(cs);
            float gdst = gv;
            if (sampled) {
                const uint32_t row = opnd_offset(I.dst, e) >> 3;
                const float zb = od + gv;
                const float eps = K.noise ? K.noise[(size_t)row * K.n_local + T.nc] : philox_normal(K, T, row);
                gloc += zb;
                gS += zb * eps;
                gdst = 0.0f;
            }
            lds_st(cd, od + gdst);
            lds_st(ca, oa + gloc * B);
            lds_st(cb, ob + gloc * A);
            lds_st(cc, oc + gloc);
            lds_st(cs, os + gS);
        } else {
            if (sampled) {
                const uint32_t doff = opnd_offset(I.dst, e), row = doff >> 3;
                const float zb = ld_adj<SM>(K, T, doff) + gv;
                const float eps = K.noise ? K.noise[(size_t)row * K.n_local + T.nc] : philox_normal(K, T, row);
                gloc += zb;
                gS += zb * eps;
            } else {
                add_adj<SM>(K, L, T, I.dst, e, gv);
            }
            add_adj<SM>(K, L, T, I.a, e, gloc * B);
            add_adj<SM>(K, L, T, I.b, e, gloc * A);
            add_adj<SM>(K, L, T, I.c, e, gloc);
            add_adj<SM>(K, L, T, I.s, e, gS);
        }
    } else if (op == BSVI_OP_BIN) {
        const float a = ld_opnd<SM>(K, T, I.a, e), b = ld_opnd<SM>(K, T, I.b, e);
        const uint32_t d = opnd_offset(I.dst, e);
        const float g = ld_adj<SM>(K, T, d);
        float ga = 0.0f, gb = 0.0f;
        switch (flags) {
        case BSVI_B_ADD: ga = g; gb = g; break;
        case BSVI_B_SUB: ga = g; gb = -g; break;
        case BSVI_B_MUL: ga = g * b; gb = g * a; break;
        case BSVI_B_DIV: ga = g / b; gb = -ga * ld_slot<SM>(K, T, d); break;
        case BSVI_B_POW:
            if (GEN) { ga = g * b * pow_ff(a, b - 1.0f); gb = (g == 0.0f) ? 0.0f : g * ld_slot<SM>(K, T, d) * logf(a); }
            break;
        default: break;
        }
        add_adj<SM>(K, L, T, I.a, e, ga);
        add_adj<SM>(K, L, T, I.b, e, gb);
    } else if (op == BSVI_OP_UN) {
        const float a = ld_opnd<SM>(K, T, I.a, e);
        const uint32_t d = opnd_offset(I.dst, e);
        const float g = ld_adj<SM>(K, T, d);
        add_adj<SM>(K, L, T, I.a, e, g * unop_grad<GEN>(flags, a, ld_slot<SM>(K, T, d), __uint_as_float(I.imm0)));
    } else if (GEN && op == BSVI_OP_NODE) {
        const int dist = (int)((I.w0 >> 16) & 0xFFu);
        const float p0 = ld_opnd<SM>(K, T, I.a, e), p1 = ld_opnd<SM>(K, T, I.b, e);
        const float v = ld_opnd<SM>(K, T, I.dst, e);
        float gv = 0.0f, g0 = 0.0f, g1 = 0.0f;
        if (flags & (BSVI_F_LOGP | BSVI_F_WF)) {
            const float gw = __uint_as_float(I.imm0) * T.mask + ((flags & BSVI_F_WF) ? T.fweight : 0.0f);
            const float4 r = logp_bwd_generic(dist, v, p0, p1, gw);
            gv += r.x; g0 += r.y; g1 += r.z;
        }
        if (flags & BSVI_F_ENT) {
            const float2 r = entropy_bwd_generic(dist, p0, p1, __uint_as_float(I.imm1) * T.mask);
            g0 += r.x; g1 += r.y;
        }
        if (flags & BSVI_F_SAMPLE) {
            const uint32_t doff = opnd_offset(I.dst, e), row = doff >> 3;
            const float zb = ld_adj<SM>(K, T, doff) + gv;
            float noise = v;
            if (dist == BSVI_DIST_LOGNORMAL || dist == BSVI_DIST_CAUCHY || dist == BSVI_DIST_LAPLACE)
                noise = K.noise ? K.noise[(size_t)row * K.n_local + T.nc] : philox_noise_again(philox_key(K, T), dist, row);
            const float2 r = sample_bwd_generic(dist, v, p0, p1, noise, zb);
            g0 += r.x; g1 += r.y;
        } else {
            add_adj<SM>(K, L, T, I.dst, e, gv);
        }
        add_adj<SM>(K, L, T, I.a, e, g0);
        add_adj<SM>(K, L, T, I.b, e, g1);
    }
}

// ---------------------------------------------------------------------------------------
// The workgroup body.  On exit (after the trailing barrier):
//   g_lds[uadj + k * n_waves] = workgroup sum of d(sum_s v_s)/dU[k]     (k < n_uniform_grad)
//   g_lds[red + 0] = workgroup sum of the per-sample estimator value, [red + 1] = #non-finite
// ---------------------------------------------------------------------------------------
template <int SM>
__device__ __forceinline__ Lay make_layout(const KParams& K, uint32_t n_waves) {
    Lay L;
    L.n_waves = n_waves;
    L.nthreads = n_waves * 64;
    L.ugrad_bytes = K.n_uniform_grad * 4;
    L.obs = K.n_uniform;
    L.uadj = K.n_uniform + K.n_obs;
    L.red = L.uadj + K.n_uniform_grad * n_waves;
    L.rows = L.red + 4 * n_waves;
    L.uacc = 2 * K.n_slots * 4;
    L.dummy = L.uacc + L.ugrad_bytes;
    L.row_words = (SM == SM_LACC) ? ((2 * K.n_slots + K.n_uniform_grad + 1 + (K.stash ? K.n_noise : 0u)) | 1u)
                                  : (2 * K.n_slots + 1);
    L.lpw = K.lpw;
    L.rpw = K.lpw + (K.lpw < 64u ? 1u : 0u);
    L.aux = (L.rows + n_waves * L.rpw * L.row_words + 3u) & ~3u;   // 16-byte aligned: entries are read as uint4
    return L;
}

template <int SM, bool OUT, bool GEN>
__device__ __forceinline__ void elbo_block(const KParams& K, const Lay& L, uint32_t block_first_sample, bool copy_aux = true) {
    // pre-resolved NAFF handlers (Aux table) apply.  Not in the generic build: its out-of-line distribution
    // code already sits at the 128-VGPR limit of a 1024-thread workgroup, and the two entry buffers would spill
    constexpr bool FASTK = (SM == SM_LACC) && !OUT && !GEN;
    const uint32_t tid = threadIdx.x, nthreads = L.nthreads;
    if (lds_base_of_dynamic_array() != 0u) __builtin_trap();      // lds_ld / lds_st use absolute addresses
    if (FASTK && copy_aux) {
        // (the persistent trainer copies once: nothing else writes this region)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(K.aux);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&g_lds[L.aux]);
        for (uint32_t i = tid; i < K.n_code * kAuxWords; i += nthreads) dst[i] = src[i];
    }
#define BSVI_STAMP(i)                                                                      \
    if (OUT && K.stamps && blockIdx.x == 0 && tid == 0) {                                    \
        K.stamps[2 * (i)] = __builtin_amdgcn_s_memtime();                                    \
        K.stamps[2 * (i) + 1] = __builtin_amdgcn_s_memrealtime();                            \
    }
    BSVI_STAMP(0)
    for (uint32_t k = tid; k < K.n_uniform; k += nthreads) {
        const bsvi_uniform_entry e = K.uniform[k];
        // agent-scope load: the persistent trainer rewrites params between iterations, so the
        // read must not be served from a stale L1 line
        const float x = e.is_param ? __hip_atomic_load(&K.params[e.src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                   : K.consts[e.src];
        g_lds[k] = e.a + e.b * utransform(e.transform, x);
    }
    for (uint32_t i = tid; i < K.n_obs; i += nthreads) g_lds[L.obs + i] = K.obs[i];
    if (SM != SM_LACC)
        for (uint32_t i = tid; i < K.n_uniform_grad * L.n_waves; i += nthreads) g_lds[L.uadj + i] = 0.0f;

    Lane T;
    T.tid = tid;
    T.wave = tid >> 6;
    T.lane = tid & 63u;
    // narrow geometry (lpw < 64): only the first lpw lanes of a wave carry samples; the others run along on one
    // shared dummy row (their results are masked out everywhere), so that the rows of a wave fit LDS
    T.n = block_first_sample + T.wave * L.lpw + T.lane;
    const bool active = T.lane < L.lpw && T.n < K.n_local;
    T.nc = active ? T.n : (K.n_local - 1);
    T.mask = active ? (K.f_weight ? K.f_weight[T.nc] : 1.0f) : 0.0f;
    T.f = 0.0f;
    T.lq = 0.0f;
    T.fweight = 0.0f;
    T.nidx = K.sample_base + T.nc;
    T.cached_group = 0xFFFFFFFFu;
    T.c0 = T.c1 = T.c2 = T.c3 = 0.0f;
    T.zrow = (L.rows + (T.wave * L.rpw + min(T.lane, L.lpw)) * L.row_words) * 4u;

    for (uint32_t s = 0; s < K.n_slots; ++s) st_adj<SM>(K, T, s * 8u, 0.0f);
    if (SM == SM_LACC)
        for (uint32_t k = 0; k <= K.n_uniform_grad; ++k) lds_st(T.zrow + L.uacc + 4u * k, 0.0f);   // + dummy
    __syncthreads();
    BSVI_STAMP(1)

    // ---------------- forward sweep (sink records: value AND adjoints, see BSVI_R_SINK)
    if constexpr (FASTK) {
        // The sweep walks the Aux table (w0 of every instruction is in it); the 32-byte instruction is read
        // only on the generic path.  Two entry buffers alternate: the entry of visit i+1 is requested at the
        // start of visit i.
        uint32_t pc = 0;
        AuxRaw XA = ld_aux(L.aux, 0), XB = XA;
        auto visit = [&](const AuxRaw& Xr, AuxRaw& Xn) {
            const Aux X = aux_wait(Xr);
            const uint32_t w0 = uniform_u32(X.w0), op = w0 & 0xFFu;
            if (w0 & (kFastFlag << 24)) {
                const uint32_t npc = pc + 1;
                Xn = ld_aux(L.aux, npc < K.n_code ? npc : 0);
                if (w0 & (BSVI_R_SINK << 24)) fast_sink(T, X);
                else fast_forward(K, T, X, w0);
                pc = npc;
                return;
            }
            const bool sink = (w0 >> 24) & BSVI_R_SINK;
            if (op == BSVI_OP_REC_BEGIN) {
                const Insn I = ld_insn(K.code, pc);
                const uint32_t n = I.dst, n_elems = I.a, temp_base = I.b, n_temps = I.c;
                // a one-instruction body (an observed likelihood over its datapoints: the common record) is
                // fetched once, not once per element and sweep — the fetch is an exposed scalar-cache round trip
                const Insn J1 = ld_insn(K.code, pc + 1);
                for (uint32_t e = 0; e < n_elems; ++e) {
                    for (uint32_t j = 1; j <= n; ++j) {
                        Insn J = J1;
                        if (n != 1) J = ld_insn(K.code, pc + j);
                        exec_forward<SM, OUT, true, GEN>(K, L, T, J, e);
                    }
                    if (sink) {
                        for (uint32_t t = 0; t < n_temps; ++t) st_adj<SM>(K, T, (temp_base + t) * 8u, 0.0f);
                        for (uint32_t j = n; j >= 1; --j) {
                            Insn J = J1;
                            if (n != 1) J = ld_insn(K.code, pc + j);
                            exec_backward<SM, GEN>(K, L, T, J, e);
                        }
                    }
                }
                pc += n + 2;
                Xn = ld_aux(L.aux, pc < K.n_code ? pc : 0);
            } else {
                const Insn I = ld_insn(K.code, pc);
                const uint32_t npc = pc + 1;
                Xn = ld_aux(L.aux, npc < K.n_code ? npc : 0);
                if (sink && op == BSVI_OP_NAFF) {
                    naff_sink<SM>(K, L, T, I, 0);
                } else {
                    exec_forward<SM, OUT, true, GEN>(K, L, T, I, 0);
                    if (sink) exec_backward<SM, GEN>(K, L, T, I, 0);
                }
                pc = npc;
            }
        };
        while (pc < K.n_code) {
            visit(XA, XB);
            if (pc >= K.n_code) break;
            visit(XB, XA);
        }
    } else {
        uint32_t pc = 0;
        Insn I = ld_insn(K.code, 0);
        while (pc < K.n_code) {
            const uint32_t op = I.w0 & 0xFFu;
            const bool sink = (I.w0 >> 24) & BSVI_R_SINK;
            if (op == BSVI_OP_REC_BEGIN) {
                const uint32_t n = I.dst, n_elems = I.a, temp_base = I.b, n_temps = I.c;
                // a one-instruction body (an observed likelihood over its datapoints: the common record) is
                // fetched once, not once per element and sweep — the fetch is an exposed scalar-cache round trip
                const Insn J1 = ld_insn(K.code, pc + 1);
                for (uint32_t e = 0; e < n_elems; ++e) {
                    for (uint32_t j = 1; j <= n; ++j) {
                        Insn J = J1;
                        if (n != 1) J = ld_insn(K.code, pc + j);
                        exec_forward<SM, OUT, true, GEN>(K, L, T, J, e);
                    }
                    if (sink) {
                        for (uint32_t t = 0; t < n_temps; ++t) st_adj<SM>(K, T, (temp_base + t) * 8u, 0.0f);
                        for (uint32_t j = n; j >= 1; --j) {
                            Insn J = J1;
                            if (n != 1) J = ld_insn(K.code, pc + j);
                            exec_backward<SM, GEN>(K, L, T, J, e);
                        }
                    }
                }
                pc += n + 2;
                I = ld_insn(K.code, pc < K.n_code ? pc : 0);
            } else {
                // fetch the next instruction while this one executes
                const uint32_t npc = pc + 1;
                const Insn nxt = ld_insn(K.code, npc < K.n_code ? npc : 0);
                if (sink && op == BSVI_OP_NAFF) {
                    naff_sink<SM>(K, L, T, I, 0);
                } else {
                    exec_forward<SM, OUT, true, GEN>(K, L, T, I, 0);
                    if (sink) exec_backward<SM, GEN>(K, L, T, I, 0);
                }
                I = nxt;
                pc = npc;
            }
        }
    }
    BSVI_STAMP(2)
    const float value = (K.estimator == BSVI_EST_BLACKBOX) ? (T.lq * T.f + T.f) : T.f;
    T.fweight = K.q_weight ? (active ? K.q_weight[T.nc] : 0.0f) : T.f * T.mask;
    if (OUT && K.fvalue_out && active) {
        K.fvalue_out[T.n] = T.f;
        K.fvalue_out[(size_t)K.n_local + T.n] = T.lq;
    }

    // ---------------- reverse sweep: the posterior's sampling chain and the derived values
    if constexpr (FASTK) {
        uint32_t pc = K.n_code;
        AuxRaw XA = ld_aux(L.aux, pc - 1), XB = XA;
        auto visit = [&](const AuxRaw& Xr, AuxRaw& Xn) {
            const Aux X = aux_wait(Xr);
            const uint32_t w0 = uniform_u32(X.w0), op = w0 & 0xFFu;
            if ((w0 & ((kFastFlag | BSVI_R_SINK) << 24)) == (kFastFlag << 24)) {
                const uint32_t npc = pc - 1;
                Xn = ld_aux(L.aux, npc > 0 ? npc - 1 : 0);
                fast_backward(K, T, X, w0);
                pc = npc;
                return;
            }
            const bool sink = (w0 >> 24) & BSVI_R_SINK;
            if (op == BSVI_OP_REC_END) {
                const Insn I = ld_insn(K.code, pc - 1);
                const uint32_t n = I.dst, n_elems = I.a, temp_base = I.b, n_temps = I.c;
                const uint32_t first = pc - 1 - n;          // index of the first body instruction
                if (!sink) {
                    const Insn J1 = ld_insn(K.code, first);
                    for (uint32_t e = n_elems; e-- > 0;) {
                        // temps are shared by all records: re-materialise this record's, clear their adjoints
                        for (uint32_t j = 0; j < n; ++j) {
                            Insn J = J1;
                            if (n != 1) J = ld_insn(K.code, first + j);
                            exec_forward<SM, false, false, GEN>(K, L, T, J, e);
                        }
                        for (uint32_t t = 0; t < n_temps; ++t) st_adj<SM>(K, T, (temp_base + t) * 8u, 0.0f);
                        for (uint32_t j = n; j-- > 0;) {
                            Insn J = J1;
                            if (n != 1) J = ld_insn(K.code, first + j);
                            exec_backward<SM, GEN>(K, L, T, J, e);
                        }
                    }
                }
                pc = first - 1;                              // skip the REC_BEGIN bracket too
                Xn = ld_aux(L.aux, pc > 0 ? pc - 1 : 0);
            } else {
                const uint32_t npc = pc - 1;
                Xn = ld_aux(L.aux, npc > 0 ? npc - 1 : 0);
                if (!sink) {
                    const Insn I = ld_insn(K.code, pc - 1);
                    exec_backward<SM, GEN>(K, L, T, I, 0);
                }
                pc = npc;
            }
        };
        while (pc > 0) {
            visit(XA, XB);
            if (pc == 0) break;
            visit(XB, XA);
        }
    } else {
        uint32_t pc = K.n_code;
        Insn I = ld_insn(K.code, pc - 1);
        while (pc > 0) {
            const uint32_t op = I.w0 & 0xFFu;
            const bool sink = (I.w0 >> 24) & BSVI_R_SINK;
            if (op == BSVI_OP_REC_END) {
                const uint32_t n = I.dst, n_elems = I.a, temp_base = I.b, n_temps = I.c;
                const uint32_t first = pc - 1 - n;          // index of the first body instruction
                if (!sink) {
                    const Insn J1 = ld_insn(K.code, first);
                    for (uint32_t e = n_elems; e-- > 0;) {
                        // temps are shared by all records: re-materialise this record's, clear their adjoints
                        for (uint32_t j = 0; j < n; ++j) {
                            Insn J = J1;
                            if (n != 1) J = ld_insn(K.code, first + j);
                            exec_forward<SM, false, false, GEN>(K, L, T, J, e);
                        }
                        for (uint32_t t = 0; t < n_temps; ++t) st_adj<SM>(K, T, (temp_base + t) * 8u, 0.0f);
                        for (uint32_t j = n; j-- > 0;) {
                            Insn J = J1;
                            if (n != 1) J = ld_insn(K.code, first + j);
                            exec_backward<SM, GEN>(K, L, T, J, e);
                        }
                    }
                }
                pc = first - 1;                              // skip the REC_BEGIN bracket too
                I = ld_insn(K.code, pc > 0 ? pc - 1 : 0);
            } else {
                const uint32_t npc = pc - 1;
                const Insn nxt = ld_insn(K.code, npc > 0 ? npc - 1 : 0);
                if (!sink) exec_backward<SM, GEN>(K, L, T, I, 0);
                I = nxt;
                pc = npc;
            }
        }
    }
    BSVI_STAMP(3)

    // ---------------- workgroup reduction (fixed order)
    const float vsum = wave_sum(active ? value : 0.0f);
    const float nonfinite = wave_sum((active && !isfinite(value)) ? 1.0f : 0.0f);
    if (T.lane == 0) {
        g_lds[L.red + 2 + 2 * T.wave] = vsum;
        g_lds[L.red + 3 + 2 * T.wave] = nonfinite;
    }
    __syncthreads();
    if (SM == SM_LACC) {
        // sum the per-lane accumulators of every entry over the workgroup's samples, in two
        // fixed-order stages: n_waves partial sums per entry (one per 64-lane group), then their sum
        const uint32_t row_bytes = L.row_words * 4u, cell0 = L.rows * 4u + L.uacc;
        for (uint32_t i = tid; i < K.n_uniform_grad * L.n_waves; i += nthreads) {
            const uint32_t k = i / L.n_waves, wv = i - k * L.n_waves;
            float s = 0.0f;
            for (uint32_t l = wv * L.rpw; l < wv * L.rpw + L.lpw; ++l) s += lds_ld(cell0 + 4u * k + l * row_bytes);
            g_lds[L.uadj + i] = s;
        }
        __syncthreads();
    }
    for (uint32_t k = tid; k < K.n_uniform_grad; k += nthreads) {
        float s = 0.0f;
        for (uint32_t wv = 0; wv < L.n_waves; ++wv) s += g_lds[L.uadj + k * L.n_waves + wv];
        g_lds[L.uadj + k * L.n_waves] = s;
    }
    if (tid == 0) {
        float s = 0.0f, c = 0.0f;
        for (uint32_t wv = 0; wv < L.n_waves; ++wv) { s += g_lds[L.red + 2 + 2 * wv]; c += g_lds[L.red + 3 + 2 * wv]; }
        g_lds[L.red] = s;
        g_lds[L.red + 1] = c;
    }
    __syncthreads();
    BSVI_STAMP(4)
}

template <int SM, bool OUT, bool GEN>
__global__ void __launch_bounds__(1024) elbo_kernel(const KParams K_in) {
    KParams K = K_in;
    if (K.offset_dev) {
        const unsigned long long o = *K.offset_dev + (((unsigned long long)K.offset_hi << 32) | K.offset_lo);
        K.offset_lo = (uint32_t)o;
        K.offset_hi = (uint32_t)(o >> 32);
    }
    if (K.n_shares > 1) {
        // every share samples the posterior but evaluates only its part of the model's log-prob records: value and
        // adjoints are linear in them, so the rows of partial sums add up in reduce_kernel like those of sample groups
        KParams S = K;
        const uint32_t share = blockIdx.x % K.n_shares;
        S.code = K.share_code[share];
        S.aux = K.share_aux[share];
        S.n_code = K.share_n_code[share];
        const Lay LS = make_layout<SM>(S, blockDim.x >> 6);
        elbo_block<SM, OUT, GEN>(S, LS, (blockIdx.x / K.n_shares) * (blockDim.x >> 6) * K.lpw);
        float* part = K.partials + (size_t)blockIdx.x * (2 + K.n_uniform_grad);
        if (threadIdx.x == 0) { part[0] = g_lds[LS.red]; part[1] = g_lds[LS.red + 1]; }
        for (uint32_t k = threadIdx.x; k < K.n_uniform_grad; k += blockDim.x) part[2 + k] = g_lds[LS.uadj + k * LS.n_waves];
        return;
    }
    const Lay L = make_layout<SM>(K, blockDim.x >> 6);
    elbo_block<SM, OUT, GEN>(K, L, blockIdx.x * (blockDim.x >> 6) * K.lpw);
    if (K.fuse_out) {
        // the only workgroup: what reduce_kernel would do with its one block of partials (sums, not yet scaled)
        float* out = K.fuse_out;
        if (threadIdx.x == 0) { out[0] = g_lds[L.red]; out[1] = g_lds[L.red + 1]; }
        for (uint32_t i = threadIdx.x; i < K.n_params; i += blockDim.x) {
            float gsum = 0.0f;
            const float theta = K.params[i];
            for (uint32_t j = K.pu_ptr[i]; j < K.pu_ptr[i + 1]; ++j) {
                const uint32_t k = K.pu_idx[j];
                const bsvi_uniform_entry e = K.uniform[k];
                gsum += g_lds[L.uadj + k * L.n_waves] * (e.b * utransform_grad(e.transform, theta));
            }
            out[BSVI_OUT_HEADER + i] = gsum;
        }
        return;
    }
    float* part = K.partials + (size_t)blockIdx.x * (2 + K.n_uniform_grad);
    if (threadIdx.x == 0) { part[0] = g_lds[L.red]; part[1] = g_lds[L.red + 1]; }
    for (uint32_t k = threadIdx.x; k < K.n_uniform_grad; k += blockDim.x) part[2 + k] = g_lds[L.uadj + k * L.n_waves];
}

struct RParams {
    const bsvi_uniform_entry* uniform;
    const uint32_t* pu_ptr;
    const uint32_t* pu_idx;
    const float* partials;
    float* params;          // non-const: the fused step updates it
    float* out;
    float* state;
    const uint8_t* active_mask;
    float* loss_slot;       // optional: where to log the loss of this iteration
    float* finite_slot;
    uint32_t n_uniform_grad, n_params, n_blocks, n_global;
    uint32_t do_finalize, do_step;
    bsvi_opt_cfg cfg;
};

// Many workgroups (throughput shards): column k of the [n_blocks][2 + nUg] partial sums is added by workgroup k
// in a fixed order (256 interleaved slices, then a tree), into one extra row behind the table; reduce_kernel
// then sees a single block.  Without it the single reduce workgroup walked 1024 rows serially: 310 us of a
// 460 us step at number_samples = 262144.
__global__ void __launch_bounds__(256) column_sum_kernel(const float* partials, float* row_out, uint32_t n_blocks, uint32_t stride) {
    __shared__ float red[4];
    const uint32_t k = blockIdx.x, tid = threadIdx.x;
    float s = 0.0f;
    for (uint32_t b = tid; b < n_blocks; b += 256) s += partials[(size_t)b * stride + k];
    const float w = wave_sum(s);
    if ((tid & 63u) == 0) red[tid >> 6] = w;
    __syncthreads();
    if (tid == 0) row_out[k] = (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup: partial sums -> gradient sums (-> loss / grads -> optimizer step).
__global__ void reduce_kernel(const RParams R) {
    float* usum = g_lds;                     // [n_uniform_grad]
    __shared__ float hdr[4];
    const uint32_t stride = 2 + R.n_uniform_grad;
    for (uint32_t k = threadIdx.x; k < R.n_uniform_grad; k += blockDim.x) {
        float s = 0.0f;
        for (uint32_t b = 0; b < R.n_blocks; ++b) s += R.partials[(size_t)b * stride + 2 + k];
        usum[k] = s;
    }
    if (threadIdx.x == 0) {
        float s = 0.0f, c = 0.0f;
        for (uint32_t b = 0; b < R.n_blocks; ++b) { s += R.partials[(size_t)b * stride]; c += R.partials[(size_t)b * stride + 1]; }
        const float loss = -s / (float)R.n_global;
        const float finite = isfinite(loss) ? 1.0f : 0.0f;
        hdr[0] = s; hdr[1] = c; hdr[2] = loss; hdr[3] = finite;
        R.out[0] = s;
        R.out[1] = c;
        if (R.do_finalize) {
            R.out[2] = loss;
            R.out[3] = finite;
            if (R.loss_slot) *R.loss_slot = loss;
            if (R.finite_slot) *R.finite_slot = finite;
        }
    }
    __syncthreads();
    const float scale = R.do_finalize ? -1.0f / (float)R.n_global : 1.0f;
    for (uint32_t i = threadIdx.x; i < R.n_params; i += blockDim.x) {
        float gsum = 0.0f;
        const float theta = R.params[i];
        for (uint32_t j = R.pu_ptr[i]; j < R.pu_ptr[i + 1]; ++j) {
            const uint32_t k = R.pu_idx[j];
            const bsvi_uniform_entry e = R.uniform[k];
            gsum += usum[k] * (e.b * utransform_grad(e.transform, theta));
        }
        const float grad = gsum * scale;
        R.out[BSVI_OUT_HEADER + i] = grad;
        if (R.do_step && hdr[3] != 0.0f && R.active_mask[i]) optimizer_update(R.cfg, R.params, R.state, R.n_params, i, grad);
    }
}

__global__ void finalize_kernel(float* out, uint32_t n_params, uint32_t n_global) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float scale = -1.0f / (float)n_global;
    if (i == 0) {
        const float loss = out[0] * scale;
        out[2] = loss;
        out[3] = isfinite(loss) ? 1.0f : 0.0f;
    }
    if (i < n_params) out[BSVI_OUT_HEADER + i] *= scale;
}

__global__ void optimizer_kernel(const bsvi_opt_cfg cfg, float* params, const float* out, float* state,
                                 const uint8_t* active_mask, uint32_t n_params) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_params || out[3] == 0.0f || !active_mask[i]) return;
    optimizer_update(cfg, params, state, n_params, i, out[BSVI_OUT_HEADER + i]);
}

// finalize + finite flag + optimizer step + loss log in ONE launch: what follows the all-reduce of the
// output block on the multi-GPU path (four launches otherwise, in a step that is latency-bound)
__global__ void finalize_step_kernel(const bsvi_opt_cfg cfg, float* params, float* out, float* state,
                                     const uint8_t* active_mask, uint32_t n_params, uint32_t n_global,
                                     float* loss_slot, float* finite_slot) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float scale = -1.0f / (float)n_global;
    const float loss = out[0] * scale;
    const float finite = isfinite(loss) ? 1.0f : 0.0f;
    if (i < n_params) {
        const float grad = out[BSVI_OUT_HEADER + i] * scale;
        out[BSVI_OUT_HEADER + i] = grad;
        if (finite != 0.0f && active_mask[i]) optimizer_update(cfg, params, state, n_params, i, grad);
    }
    // (every thread has read out[0] before anyone overwrites the header: one workgroup per 256 parameters, and
    //  the header words written below are not read above)
    if (i == 0) {
        out[2] = loss;
        out[3] = finite;
        if (loss_slot) *loss_slot = loss;
        if (finite_slot) *finite_slot = finite;
    }
}

// The same for a step sequence that is captured once in a HIP graph and replayed: nothing in the launch may change
// between replays, so the iteration number lives in device memory.  counters[0] = Philox offset of the iteration
// (read by bsvi_elbo_fwd_bwd through bsvi_elbo_args::offset_dev), counters[1] = index of the iteration within the run:
// picks the loss / finite slot and, with pretraining_iterations, the mask; both are advanced here.
__global__ void finalize_step_counted_kernel(const bsvi_opt_cfg cfg, float* params, float* out, float* state,
                                             const uint8_t* mask_all, const uint8_t* mask_first, uint32_t pretraining,
                                             uint32_t n_params, uint32_t n_global, float* loss_curve, float* finite_curve,
                                             unsigned long long* counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long it = counters[1];
    const float scale = -1.0f / (float)n_global;
    const float loss = out[0] * scale;
    const float finite = isfinite(loss) ? 1.0f : 0.0f;
    const uint8_t* mask = (it > pretraining) ? mask_all : mask_first;
    if (i < n_params) {
        const float grad = out[BSVI_OUT_HEADER + i] * scale;
        out[BSVI_OUT_HEADER + i] = grad;
        if (finite != 0.0f && mask[i]) optimizer_update(cfg, params, state, n_params, i, grad);
    }
    // one workgroup covers every parameter of the launch (checked by the host): the barrier orders every thread's
    // reads of out[0] and counters[1] before thread 0 rewrites them
    __syncthreads();
    if (i == 0) {
        out[2] = loss;
        out[3] = finite;
        if (loss_curve) loss_curve[it] = loss;
        if (finite_curve) finite_curve[it] = finite;
        counters[0] += 1ull;
        counters[1] = it + 1ull;
    }
}

// ---------------------------------------------------------------------------------------
// Persistent trainer: the whole loop of brancher/inference.py:95-108 in one launch, for
// sample counts that fit one workgroup.  Every iteration: ELBO fwd+bwd, chain rule,
// finalize, finite check, optimizer step, loss log — nothing leaves the CU.
// ---------------------------------------------------------------------------------------
struct PParams {
    KParams K;
    RParams R;
    uint32_t n_iterations;
    float* loss_curve;
    float* finite_curve;
    const uint8_t* active_mask_first;   // mask used while iteration <= pretraining_iterations
    uint32_t pretraining_iterations;
    // persistent_multi_kernel: one wave per workgroup, one workgroup per CU
    float* mw_xchg;          // [2][n_wg][2 + n_uniform_grad] partial sums, double-buffered over iterations
    uint32_t* mw_counter;    // arrivals, monotonic over the launch
    float* mw_params;        // private parameter copies of workgroups 1.. (workgroup 0 owns the caller's buffer)
    float* mw_state;         // private optimizer state copies
    uint32_t mw_stride;      // floats between two private parameter copies
    // shares of the program (bsvi_train_persistent_split): workgroup w runs share w % n_shares on sample wave
    // w / n_shares; every share samples the posterior but evaluates only its part of the model's log-prob records
    uint32_t n_shares;       // 0 / 1: the whole program
    const uint4* share_code[8];
    const uint4* share_aux[8];
    uint32_t share_n_code[8];
};

// Scalar-register discipline: PParams is ~90 dwords of kernel arguments.  Read as `P.x` they are all
// loaded at entry and stay live through the sweeps, which then spill scalars around every instruction
// visit.  So each phase re-reads what it needs from the kernarg segment (scalar loads through a
// pointer the optimizer cannot see through), and nothing but the sweeps' own state is live in them.
template <int SM, bool GEN>
__global__ void __launch_bounds__(1024) persistent_kernel(const PParams P_unused) {
    (void)P_unused;
    const uint32_t n_waves = blockDim.x >> 6;
    const BSVI_CONST_AS char* ka = (const BSVI_CONST_AS char*)__builtin_amdgcn_kernarg_segment_ptr();
#if defined(__HIP_DEVICE_COMPILE__)
#define BSVI_RELOAD_ARGS() asm volatile("" : "+s"(ka))
#else
#define BSVI_RELOAD_ARGS()
#endif
    const uint32_t n_iterations = ((const BSVI_CONST_AS PParams*)ka)->n_iterations;
    for (uint32_t it = 0; it < n_iterations; ++it) {
        Lay L;
        {
            BSVI_RELOAD_ARGS();
            KParams K = ((const BSVI_CONST_AS PParams*)ka)->K;
            // one iteration = one Philox offset; a given-noise sequence is laid out [it][row][N]
            const uint32_t lo = K.offset_lo + it;
            K.offset_hi += (lo < K.offset_lo) ? 1u : 0u;
            K.offset_lo = lo;
            if (K.noise) K.noise += (size_t)it * K.n_noise * K.n_local;
            L = make_layout<SM>(K, n_waves);
            elbo_block<SM, false, GEN>(K, L, 0, it == 0);
        }
        BSVI_RELOAD_ARGS();
        const BSVI_CONST_AS PParams* P = (const BSVI_CONST_AS PParams*)ka;
        const uint32_t n_global = P->K.n_global;
        if (threadIdx.x == 0) {
            const float vsum = g_lds[L.red], bad = g_lds[L.red + 1];
            const float loss = -vsum / (float)n_global;
            const float finite = isfinite(loss) ? 1.0f : 0.0f;
            g_lds[L.red + 2] = finite;     // red[2..] is free once elbo_block has returned
            P->loss_curve[it] = loss;
            P->finite_curve[it] = finite;
            float* out = P->R.out;
            out[0] = vsum; out[1] = bad; out[2] = loss; out[3] = finite;
        }
        __syncthreads();
        const float scale = -1.0f / (float)n_global;
        float* const params = P->R.params;
        float* const state = P->R.state;
        float* const out = P->R.out;
        const uint32_t* const pu_ptr = P->R.pu_ptr;
        const uint32_t* const pu_idx = P->R.pu_idx;
        const bsvi_uniform_entry* const uniform = P->R.uniform;
        const uint32_t n_params = P->R.n_params;
        const uint8_t* mask = (it > P->pretraining_iterations) ? P->R.active_mask : P->active_mask_first;
        for (uint32_t i = threadIdx.x; i < n_params; i += blockDim.x) {
            float gsum = 0.0f;
            const float theta = __hip_atomic_load(&params[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t j = pu_ptr[i]; j < pu_ptr[i + 1]; ++j) {
                const uint32_t k = pu_idx[j];
                const bsvi_uniform_entry e = uniform[k];
                gsum += g_lds[L.uadj + k * n_waves] * (e.b * utransform_grad(e.transform, theta));
            }
            const float grad = gsum * scale;
            out[BSVI_OUT_HEADER + i] = grad;
            if (g_lds[L.red + 2] != 0.0f && mask[i]) {
                const bsvi_opt_cfg cfg = P->R.cfg;
                optimizer_update(cfg, params, state, n_params, i, grad);
            }
        }
        // The parameters are written and re-read by this one workgroup only: the stores are
        // drained by the barrier below (s_waitcnt vmcnt(0) + s_barrier) and the next prologue
        // re-reads them with agent-scope loads that bypass the L1.
        __syncthreads();
    }
#undef BSVI_RELOAD_ARGS
}

// ---------------------------------------------------------------------------------------
// Persistent trainer over SEVERAL workgroups of one wave each.  A 300-sample shard is five waves: in one
// workgroup two of them share a SIMD, and that SIMD — issue-bound, ~4.7 cycles per instruction per wave — sets the
// iteration time.  Here every wave has a CU (and a SIMD) to itself.  Per iteration ONE exchange: each workgroup
// publishes its [2 + n_uniform_grad] partial sums, all meet at a counter (agent-scope release / acquire), and every
// workgroup then adds the partials in the same fixed order and applies the same chain rule and optimizer update to
// its OWN copy of the parameters and the optimizer state — bitwise the same everywhere, so no second exchange is
// needed.  Workgroup 0 owns the caller's buffers and writes the loss curve and the gradient block.
// ---------------------------------------------------------------------------------------
template <int SM, bool GEN>
__global__ void __launch_bounds__(1024) persistent_multi_kernel(const PParams P_unused) {
    (void)P_unused;
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t wg = blockIdx.x, n_wg = gridDim.x;
    const BSVI_CONST_AS char* ka = (const BSVI_CONST_AS char*)__builtin_amdgcn_kernarg_segment_ptr();
#if defined(__HIP_DEVICE_COMPILE__)
#define BSVI_RELOAD_ARGS() asm volatile("" : "+s"(ka))
#else
#define BSVI_RELOAD_ARGS()
#endif
    const uint32_t n_iterations = ((const BSVI_CONST_AS PParams*)ka)->n_iterations;
    if (wg > 0) {      // private copies of the parameters and of the optimizer state
        const BSVI_CONST_AS PParams* P = (const BSVI_CONST_AS PParams*)ka;
        const uint32_t n_params = P->R.n_params;
        float* mp = P->mw_params + (size_t)(wg - 1) * P->mw_stride;
        float* ms = P->mw_state + (size_t)(wg - 1) * 4 * P->mw_stride;
        for (uint32_t i = threadIdx.x; i < n_params; i += blockDim.x) mp[i] = P->R.params[i];
        for (uint32_t i = threadIdx.x; i < 4 * n_params; i += blockDim.x) ms[i] = P->R.state[i];
        __syncthreads();
    }
    for (uint32_t it = 0; it < n_iterations; ++it) {
        Lay L;
        {
            BSVI_RELOAD_ARGS();
            KParams K = ((const BSVI_CONST_AS PParams*)ka)->K;
            const uint32_t lo = K.offset_lo + it;
            K.offset_hi += (lo < K.offset_lo) ? 1u : 0u;
            K.offset_lo = lo;
            if (K.noise) K.noise += (size_t)it * K.n_noise * K.n_local;
            const BSVI_CONST_AS PParams* P0 = (const BSVI_CONST_AS PParams*)ka;
            if (wg > 0) K.params = P0->mw_params + (size_t)(wg - 1) * P0->mw_stride;
            uint32_t sample_wave = wg;
            if (P0->n_shares > 1) {
                const uint32_t share = wg % P0->n_shares;
                sample_wave = wg / P0->n_shares;
                K.code = P0->share_code[share];
                K.aux = P0->share_aux[share];
                K.n_code = P0->share_n_code[share];
            }
            L = make_layout<SM>(K, n_waves);
            elbo_block<SM, false, GEN>(K, L, sample_wave * n_waves * K.lpw, it == 0);
        }
        BSVI_RELOAD_ARGS();
        const BSVI_CONST_AS PParams* P = (const BSVI_CONST_AS PParams*)ka;
        const uint32_t n_ug = P->R.n_uniform_grad, stride = 2 + n_ug;
        {   // publish this workgroup's sums, meet the others, add everybody's in workgroup order
            float* slot = P->mw_xchg + (size_t)(it & 1u) * n_wg * stride;
            float* mine = slot + (size_t)wg * stride;
            for (uint32_t k = threadIdx.x; k < n_ug; k += blockDim.x)
                __hip_atomic_store(&mine[2 + k], g_lds[L.uadj + k * n_waves], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x == 0) {
                __hip_atomic_store(&mine[0], g_lds[L.red], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&mine[1], g_lds[L.red + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();                       // every wave's stores are issued and drained (vmcnt(0) + barrier)
            if (threadIdx.x == 0) {
                // counter[0]: arrivals; counter[1]: abort flag.  The launch is not cooperative: should a workgroup not be
                // resident (the host checks that there is a CU for each, but other streams / processes can hold CUs), the
                // others must neither spin forever nor take the process down — the first to give up raises the flag,
                // everybody leaves the loop, and the rest of the loss curve reads NaN / not finite.
                uint32_t* counter = P->mw_counter;
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t target = (it + 1u) * n_wg;
                uint32_t polls = 0;
                float aborted = 0.0f;
                while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++polls > (1u << 24) || __hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        __hip_atomic_store(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        aborted = 1.0f;
                        break;
                    }
                }
                g_lds[L.red + 3] = aborted;
            }
            __syncthreads();
            if (g_lds[L.red + 3] != 0.0f) {
                if (wg == 0)
                    for (uint32_t j = it + threadIdx.x; j < n_iterations; j += blockDim.x) {
                        P->loss_curve[j] = __builtin_nanf("");
                        P->finite_curve[j] = 0.0f;
                    }
                return;
            }
            for (uint32_t k = threadIdx.x; k < n_ug; k += blockDim.x) {
                float s = 0.0f;
                for (uint32_t b = 0; b < n_wg; ++b)
                    s += __hip_atomic_load(&slot[(size_t)b * stride + 2 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g_lds[L.uadj + k * n_waves] = s;
            }
            if (threadIdx.x == 0) {
                float s = 0.0f, c = 0.0f;
                for (uint32_t b = 0; b < n_wg; ++b) {
                    s += __hip_atomic_load(&slot[(size_t)b * stride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    c += __hip_atomic_load(&slot[(size_t)b * stride + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                g_lds[L.red] = s;
                g_lds[L.red + 1] = c;
            }
            __syncthreads();
        }
        const uint32_t n_global = P->K.n_global;
        if (threadIdx.x == 0) {
            const float vsum = g_lds[L.red], bad = g_lds[L.red + 1];
            const float loss = -vsum / (float)n_global;
            const float finite = isfinite(loss) ? 1.0f : 0.0f;
            g_lds[L.red + 2] = finite;
            if (wg == 0) {
                P->loss_curve[it] = loss;
                P->finite_curve[it] = finite;
                float* out = P->R.out;
                out[0] = vsum; out[1] = bad; out[2] = loss; out[3] = finite;
            }
        }
        __syncthreads();
        const float scale = -1.0f / (float)n_global;
        float* const params = wg == 0 ? P->R.params : P->mw_params + (size_t)(wg - 1) * P->mw_stride;
        float* const state = wg == 0 ? P->R.state : P->mw_state + (size_t)(wg - 1) * 4 * P->mw_stride;
        float* const out = P->R.out;
        const uint32_t* const pu_ptr = P->R.pu_ptr;
        const uint32_t* const pu_idx = P->R.pu_idx;
        const bsvi_uniform_entry* const uniform = P->R.uniform;
        const uint32_t n_params = P->R.n_params;
        const uint8_t* mask = (it > P->pretraining_iterations) ? P->R.active_mask : P->active_mask_first;
        for (uint32_t i = threadIdx.x; i < n_params; i += blockDim.x) {
            float gsum = 0.0f;
            const float theta = __hip_atomic_load(&params[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t j = pu_ptr[i]; j < pu_ptr[i + 1]; ++j) {
                const uint32_t k = pu_idx[j];
                const bsvi_uniform_entry e = uniform[k];
                gsum += g_lds[L.uadj + k * n_waves] * (e.b * utransform_grad(e.transform, theta));
            }
            const float grad = gsum * scale;
            if (wg == 0) out[BSVI_OUT_HEADER + i] = grad;
            if (g_lds[L.red + 2] != 0.0f && mask[i]) {
                const bsvi_opt_cfg cfg = P->R.cfg;
                optimizer_update(cfg, params, state, n_params, i, grad);
            }
        }
        __syncthreads();
    }
#undef BSVI_RELOAD_ARGS
}

// ---------------------------------------------------------------------------------------
// test hook: evaluate one node/special function elementwise (tests/test_gpu_math.py)
// ---------------------------------------------------------------------------------------
__global__ void debug_math_kernel(int fn, int dist, const float* x, const float* p0, const float* p1,
                                  float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xv = x[i], a = p0[i], b = p1[i];
    float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, r3 = 0.0f;
    switch (fn) {
    case 0: r0 = digammaf_(xv); break;
    case 1: r0 = trigammaf_(xv); break;
    case 2: r0 = dirichlet_grad_one(xv, a, b); break;
    case 3: { r0 = logp_generic(dist, xv, a, b); const float4 r = logp_bwd_generic(dist, xv, a, b, 1.0f); r1 = r.x; r2 = r.y; r3 = r.z; break; }
    case 4: { r0 = entropy_generic(dist, a, b); const float2 r = entropy_bwd_generic(dist, a, b, 1.0f); r2 = r.x; r3 = r.y; break; }
    case 5: { r0 = sample_from_noise_generic(dist, a, b, xv); const float2 r = sample_bwd_generic(dist, r0, a, b, xv, 1.0f); r2 = r.x; r3 = r.y; break; }
    case 6: r0 = lgammaf(xv); break;
    default: break;
    }
    out[i] = r0; out[n + i] = r1; out[2 * (size_t)n + i] = r2; out[3 * (size_t)n + i] = r3;
}

}  // namespace bsvi

#undef ZG
// =========================================================================================
//  C ABI
// =========================================================================================
using namespace bsvi;

static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
int bsvi_fail(int code, const std::string& msg) { return fail(code, msg); }   // for amort_kernel.hip (bsvi_internal.h)
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(BSVI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

static unsigned long long* g_debug_stamps = nullptr;
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" const char* bsvi_last_error(void) { return g_last_error.c_str(); }
extern "C" int bsvi_abi_version(void) { return BSVI_ABI_VERSION; }
extern "C" size_t bsvi_sizeof(int kind) {
    switch (kind) {
    case BSVI_SK_UNIFORM_ENTRY: return sizeof(bsvi_uniform_entry);
    case BSVI_SK_RECORD: return sizeof(bsvi_record);
    case BSVI_SK_PROGRAM_DESC: return sizeof(bsvi_program_desc);
    case BSVI_SK_ELBO_ARGS: return sizeof(bsvi_elbo_args);
    case BSVI_SK_OPT_CFG: return sizeof(bsvi_opt_cfg);
    case BSVI_SK_DENSE_DESC: return sizeof(bsvi_dense_desc);
    case BSVI_SK_DENSE_ARGS: return sizeof(bsvi_dense_args);
    case BSVI_SK_MLP_LAYER: return sizeof(bsvi_mlp_layer);
    case BSVI_SK_AMORT_DESC: return sizeof(bsvi_amort_desc);
    case BSVI_SK_AMORT_ARGS: return sizeof(bsvi_amort_args);
    case BSVI_SK_MVN_INSN: return sizeof(bsvi_mvn_insn);
    case BSVI_SK_MVN_DESC: return sizeof(bsvi_mvn_desc);
    case BSVI_SK_MVN_ARGS: return sizeof(bsvi_mvn_args);
    case BSVI_SK_BNN_LAYER: return sizeof(bsvi_bnn_layer);
    case BSVI_SK_BNN_DESC: return sizeof(bsvi_bnn_desc);
    case BSVI_SK_BNN_ARGS: return sizeof(bsvi_bnn_args);
    case BSVI_SK_REDUCE_DESC: return sizeof(bsvi_reduce_desc);
    case BSVI_SK_REDUCE_ARGS: return sizeof(bsvi_reduce_args);
    default: return 0;
    }
}

extern "C" int bsvi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int validate(const bsvi_program_desc* d) {
    if (!d) return fail(BSVI_ERR_INVALID, "null program descriptor");
    BSVI_CHECK_STRUCT(d, bsvi_program_desc);
    if (d->abi_version != BSVI_ABI_VERSION) return fail(BSVI_ERR_INVALID, "ABI version mismatch");
    if (d->n_uniform_grad > d->n_uniform) return fail(BSVI_ERR_INVALID, "n_uniform_grad > n_uniform");
    if (d->estimator > BSVI_EST_BLACKBOX) return fail(BSVI_ERR_INVALID, "unknown estimator");
    if ((d->n_uniform && !d->uniform) || (d->n_records && !d->records) || (d->n_code && !d->code))
        return fail(BSVI_ERR_INVALID, "missing program table");
    for (uint32_t k = 0; k < d->n_uniform; ++k) {
        const bsvi_uniform_entry& e = d->uniform[k];
        if (e.transform > BSVI_UT_SQUARE) return fail(BSVI_ERR_INVALID, "bad uniform transform");
        if ((k < d->n_uniform_grad) != (e.is_param != 0)) return fail(BSVI_ERR_INVALID, "uniform table not partitioned");
        if (e.src >= (e.is_param ? d->n_params : d->n_consts)) return fail(BSVI_ERR_INVALID, "uniform source out of range");
    }
    for (uint32_t r = 0; r < d->n_records; ++r) {
        const bsvi_record& R = d->records[r];
        if (R.code_begin > R.code_end || R.code_end > d->n_code) return fail(BSVI_ERR_INVALID, "record code span out of range");
        if (!R.n_elems) return fail(BSVI_ERR_INVALID, "empty record");
        if ((uint64_t)R.temp_base + R.n_temps > d->n_slots) return fail(BSVI_ERR_INVALID, "temp slots out of range");
        const uint64_t ext = R.n_elems - 1;
        for (uint32_t pc = R.code_begin; pc < R.code_end; ++pc) {
            const uint32_t* w = d->code + 8 * (size_t)pc;
            const uint32_t op = w[0] & 0xFF, flags = (w[0] >> 8) & 0xFF, dist = (w[0] >> 16) & 0xFF;
            if (op < BSVI_OP_NAFF || op > BSVI_OP_UN) return fail(BSVI_ERR_INVALID, "unknown opcode inside a record");
            if (op == BSVI_OP_BIN && flags > BSVI_B_DELTA) return fail(BSVI_ERR_INVALID, "unknown binary operator");
            if (op == BSVI_OP_UN && flags > BSVI_U_POWI) return fail(BSVI_ERR_INVALID, "unknown unary function");
            if ((op == BSVI_OP_NAFF || op == BSVI_OP_NODE) && dist >= BSVI_DIST_COUNT) return fail(BSVI_ERR_INVALID, "bad distribution id");
            if (op == BSVI_OP_NAFF && dist != BSVI_DIST_NORMAL) return fail(BSVI_ERR_INVALID, "NAFF is a Normal node");
            const int n_opnd = (op == BSVI_OP_NAFF) ? 5 : ((op == BSVI_OP_UN) ? 2 : 3);
            for (int j = 1; j <= n_opnd; ++j) {
                const uint32_t o = w[j], per_lane = o >> 31, walks = (o >> 30) & 1;
                const uint64_t last = (uint64_t)(o & 0x3FFFFFFFu) + walks * ext * (per_lane ? 8 : 4);
                const uint64_t limit = per_lane ? (uint64_t)d->n_slots * 8 : ((uint64_t)d->n_uniform + d->n_obs) * 4;
                if ((o & 3u) || (per_lane && (o & 7u))) return fail(BSVI_ERR_INVALID, "misaligned operand");
                if (last >= limit) return fail(BSVI_ERR_INVALID, "operand address out of range");
            }
            const bool writes = (op == BSVI_OP_BIN || op == BSVI_OP_UN) || (flags & BSVI_F_SAMPLE);
            if (writes && !(w[1] >> 31)) return fail(BSVI_ERR_INVALID, "destination must be a slot");
            if ((op == BSVI_OP_NAFF || op == BSVI_OP_NODE) && (flags & BSVI_F_SAMPLE) &&
                (uint64_t)((w[1] & 0x3FFFFFFFu) >> 3) + ((w[1] >> 30) & 1) * ext >= d->n_noise)
                return fail(BSVI_ERR_INVALID, "sampled slot is not a latent row");
        }
    }
    // the stream outside record bodies: plain single-element instructions and matching brackets
    {
        std::vector<uint8_t> in_body(d->n_code, 0);
        for (uint32_t r = 0; r < d->n_records; ++r)
            for (uint32_t pc = d->records[r].code_begin; pc < d->records[r].code_end; ++pc) in_body[pc] = 1;
        uint32_t r = 0;
        for (uint32_t pc = 0; pc < d->n_code;) {
            const uint32_t* w = d->code + 8 * (size_t)pc;
            const uint32_t op = w[0] & 0xFF;
            if (r >= d->n_records) return fail(BSVI_ERR_INVALID, "instructions outside any record");
            const bsvi_record& R = d->records[r];
            if (op == BSVI_OP_REC_BEGIN) {
                const uint32_t n = w[1];
                if (R.code_begin != pc + 1 || R.code_end != pc + 1 + n || pc + n + 1 >= d->n_code)
                    return fail(BSVI_ERR_INVALID, "record bracket does not match the record table");
                const uint32_t* we = d->code + 8 * (size_t)(pc + n + 1);
                if ((we[0] & 0xFF) != BSVI_OP_REC_END || memcmp(w + 1, we + 1, 16) != 0 || (w[0] >> 24) != (we[0] >> 24))
                    return fail(BSVI_ERR_INVALID, "unbalanced record brackets");
                if (w[2] != R.n_elems || w[3] != R.temp_base || w[4] != R.n_temps || ((w[0] >> 24) & 1) != (R.flags & 1))
                    return fail(BSVI_ERR_INVALID, "record bracket fields do not match the record table");
                pc += n + 2;
            } else {
                if (!in_body[pc] || R.code_begin != pc || R.code_end != pc + 1 || R.n_elems != 1 ||
                    ((w[0] >> 24) & 1) != (R.flags & 1))
                    return fail(BSVI_ERR_INVALID, "bare instruction is not a single-element record");
                pc += 1;
            }
            ++r;
        }
        if (r != d->n_records) return fail(BSVI_ERR_INVALID, "record table longer than the stream");
    }
    if (d->n_params) {
        if (!d->param_uniform_ptr || (d->n_uniform_grad && !d->param_uniform_idx)) return fail(BSVI_ERR_INVALID, "missing CSR map");
        if (d->param_uniform_ptr[d->n_params] != d->n_uniform_grad) return fail(BSVI_ERR_INVALID, "CSR map does not cover the uniform table");
    }
    return BSVI_OK;
}

extern "C" int bsvi_program_create(const bsvi_program_desc* desc, bsvi_program** out) {
    if (!out) return fail(BSVI_ERR_INVALID, "null output pointer");
    *out = nullptr;
    int rc = validate(desc);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(BSVI_ERR_NO_DEVICE, "no HIP device visible");
    bsvi_program* p = new bsvi_program();
    p->d = *desc;
    const size_t b_code = align_up((size_t)desc->n_code * 32, 256);
    const size_t b_aux = align_up((size_t)desc->n_code * kAuxWords * 4, 256);
    const size_t b_rec = align_up((size_t)desc->n_records * sizeof(bsvi_record), 256);
    const size_t b_uni = align_up((size_t)desc->n_uniform * sizeof(bsvi_uniform_entry), 256);
    const size_t b_con = align_up((size_t)desc->n_consts * 4, 256);
    const size_t b_ptr = align_up(((size_t)desc->n_params + 1) * 4, 256);
    const size_t b_idx = align_up((size_t)desc->n_uniform_grad * 4, 256);
    const size_t total = b_code + b_aux + b_rec + b_uni + b_con + b_ptr + b_idx + 256;
    std::vector<char> host(total, 0);
    size_t o = 0;
    const size_t o_code = o; if (desc->n_code) memcpy(&host[o], desc->code, (size_t)desc->n_code * 32); o += b_code;
    const size_t o_aux = o; o += b_aux;
    {
        // pre-resolve the top-level NAFF instructions (see struct Aux) and mark them in the device copy
        uint32_t* code = reinterpret_cast<uint32_t*>(&host[o_code]);
        uint32_t* aux = reinterpret_cast<uint32_t*>(&host[o_aux]);
        const uint32_t uacc = 2u * desc->n_slots * 4u, ugrad = desc->n_uniform_grad * 4u, dummy = uacc + ugrad;
        uint32_t body_left = 0;          // instructions of a bracketed record (and its closing bracket) keep the generic path
        for (uint32_t i = 0; i < desc->n_code; ++i) {
            uint32_t* w = code + 8 * (size_t)i;
            uint32_t* x = aux + kAuxWords * (size_t)i;
            const uint32_t op = w[0] & 0xFFu, flags = (w[0] >> 8) & 0xFFu, rflags = w[0] >> 24;
            x[0] = w[0]; x[1] = w[6]; x[2] = w[7];
            if (body_left) { --body_left; continue; }
            if (op == BSVI_OP_REC_BEGIN) { body_left = w[1] + 1; continue; }
            if (op != BSVI_OP_NAFF || !(rflags & BSVI_R_NOALIAS) || (flags & BSVI_F_GIVEN)) continue;
            bool ok = true;
            for (int k = 0; k < 5; ++k) ok = ok && !((w[1 + k] >> 30) & 1u);
            if (!ok) continue;
            for (int k = 0; k < 5; ++k) {
                const uint32_t o_ = w[1 + k], off = o_ & 0x3FFFFFFFu, lane = o_ >> 31;
                x[4 + k] = off;
                x[9 + k] = lane ? 0xFFFFFFFFu : 0u;
                x[14 + k] = lane ? off + 4u : (off < ugrad ? uacc + off : dummy);
            }
            x[3] = (w[1] & 0x3FFFFFFFu) >> 3;
            x[19] = dummy + 4u + 4u * x[3];          // the node's cell in the eps stash behind the dummy cell
            w[0] |= kFastFlag << 24;
            x[0] = w[0];
        }
    }
    const size_t o_rec = o; if (desc->n_records) memcpy(&host[o], desc->records, (size_t)desc->n_records * sizeof(bsvi_record)); o += b_rec;
    const size_t o_uni = o; if (desc->n_uniform) memcpy(&host[o], desc->uniform, (size_t)desc->n_uniform * sizeof(bsvi_uniform_entry)); o += b_uni;
    const size_t o_con = o; if (desc->n_consts) memcpy(&host[o], desc->consts, (size_t)desc->n_consts * 4); o += b_con;
    const size_t o_ptr = o; if (desc->n_params) memcpy(&host[o], desc->param_uniform_ptr, ((size_t)desc->n_params + 1) * 4); o += b_ptr;
    const size_t o_idx = o; if (desc->n_uniform_grad) memcpy(&host[o], desc->param_uniform_idx, (size_t)desc->n_uniform_grad * 4); o += b_idx;
    hipError_t e = hipMalloc(&p->dev_blob, total);
    if (e != hipSuccess) { delete p; return fail(BSVI_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    e = hipMemcpy(p->dev_blob, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(p->dev_blob); delete p; return fail(BSVI_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
    char* base = (char*)p->dev_blob;
    p->code = (const uint4*)(base + o_code);
    p->aux = (const uint4*)(base + o_aux);
    p->records = (const bsvi_record*)(base + o_rec);
    p->uniform = (const bsvi_uniform_entry*)(base + o_uni);
    p->consts = (const float*)(base + o_con);
    p->pu_ptr = (const uint32_t*)(base + o_ptr);
    p->pu_idx = (const uint32_t*)(base + o_idx);
    // the program-specialised kernels (specialize.cpp): sources now, hiprtc at the first launch
    {
        std::string why;
        p->spec = bsvi_spec::create(*desc, why);
        if (p->spec && bsvi_spec::upload(p->spec) != BSVI_OK) { bsvi_spec::destroy(p->spec); p->spec = nullptr; }
        if (!p->spec && getenv("BSVI_DEBUG")) fprintf(stderr, "bsvi: program not specialised: %s\n", why.c_str());
    }
    // host pointers of the descriptor are not kept
    p->d.code = nullptr; p->d.records = nullptr; p->d.uniform = nullptr; p->d.consts = nullptr;
    p->d.param_uniform_ptr = nullptr; p->d.param_uniform_idx = nullptr;
    // "generic" programs need the out-of-line paths: non-Normal nodes, pow and the rare functions
    for (uint32_t i = 0; i < desc->n_code; ++i) {
        const uint32_t* w = desc->code + 8 * (size_t)i;
        const uint32_t op = w[0] & 0xFF, sub = (w[0] >> 8) & 0xFF;
        float imm;
        memcpy(&imm, &w[6], 4);
        if (op == BSVI_OP_NODE) p->generic = true;
        if (op == BSVI_OP_BIN && sub == BSVI_B_POW) p->generic = true;
        if (op == BSVI_OP_UN) {
            const bool common = sub == BSVI_U_COPY || sub == BSVI_U_NEG || sub == BSVI_U_EXP || sub == BSVI_U_LOG ||
                                sub == BSVI_U_SQRT || sub == BSVI_U_ABS || sub == BSVI_U_SIGMOID || sub == BSVI_U_SOFTPLUS ||
                                sub == BSVI_U_RELU || sub == BSVI_U_RECIP || sub == BSVI_U_SQUARE ||
                                (sub == BSVI_U_POWI && (imm == 2.0f || imm == -1.0f || imm == 0.5f));
            if (!common) p->generic = true;
        }
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lds <= 0) lds = 65536;
    if (lds < 163840) lds = 163840;   // gfx950 has 160 KiB per CU; older runtimes under-report the opt-in limit
    // opt in to the full 160 KiB of gfx950 where the runtime allows it; never leave a sticky
    // error behind (PyTorch checks hipGetLastError after its own launches)
    const void* kernels[] = {
#define BSVI_K3(OUT_, GEN_) (const void*)elbo_kernel<SM_WSUM, OUT_, GEN_>, (const void*)elbo_kernel<SM_LACC, OUT_, GEN_>, \
                            (const void*)elbo_kernel<SM_ZG, OUT_, GEN_>
        BSVI_K3(false, false), BSVI_K3(true, false), BSVI_K3(false, true), BSVI_K3(true, true),
#undef BSVI_K3
        (const void*)persistent_kernel<SM_WSUM, false>, (const void*)persistent_kernel<SM_LACC, false>,
        (const void*)persistent_kernel<SM_ZG, false>, (const void*)persistent_kernel<SM_WSUM, true>,
        (const void*)persistent_kernel<SM_LACC, true>, (const void*)persistent_kernel<SM_ZG, true>,
        (const void*)persistent_multi_kernel<SM_LACC, false>, (const void*)persistent_multi_kernel<SM_LACC, true>,
        (const void*)reduce_kernel};
    // a kernel with static LDS cannot opt in to the full 160 KiB: leave 256 B of head-room there
    int granted = lds;
    for (const void* k : kernels) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds - 256);
            if (e == hipSuccess) {
                granted = granted < lds - 256 ? granted : lds - 256;
            } else {
                (void)hipGetLastError();
                if (getenv("BSVI_DEBUG")) fprintf(stderr, "bsvi: hipFuncSetAttribute(%d) failed: %s\n", lds, hipGetErrorString(e));
                granted = 65536;
            }
        }
    }
    p->max_lds = granted;
    *out = p;
    return BSVI_OK;
}

extern "C" void bsvi_program_destroy(bsvi_program* p) {
    if (!p) return;
    if (p->spec) bsvi_spec::destroy(p->spec);
    if (p->dev_blob) hipFree(p->dev_blob);
    delete p;
}

// ---- launch geometry ---------------------------------------------------------------------
struct Geometry {
    uint32_t n_waves = 0;   // per workgroup
    uint32_t n_blocks = 0;
    int mode = SM_WSUM;     // where slots and dU accumulators live
    bool zglobal = false;
    size_t lds_bytes = 0;
    uint32_t n_pad = 0;
    uint32_t lpw = 64;      // sample lanes per wave (lds+lane_acc only: narrower when a full wave does not fit)
    bool stash = false;     // lds+lane_acc rows also hold the sample's noise (KParams::stash)
};

static size_t lds_need(const bsvi_program* p, uint32_t n_waves, int mode, uint32_t lpw = 64, bool stash = false) {
    size_t floats = (size_t)p->d.n_uniform + (size_t)p->d.n_obs + (size_t)p->d.n_uniform_grad * n_waves + 4 * (size_t)n_waves;
    if (mode == SM_WSUM) floats += (2 * (size_t)p->d.n_slots + 1) * n_waves * 64;
    if (mode == SM_LACC) floats += ((2 * (size_t)p->d.n_slots + p->d.n_uniform_grad + 1 + (stash ? p->d.n_noise : 0)) | 1)
                                   * n_waves * (lpw + (lpw < 64 ? 1 : 0))
                                   + 4 + (size_t)p->d.n_code * kAuxWords;     // + LDS copy of the Aux table
    return floats * 4 + 64;
}

// Policy: keep samples in LDS whenever they fit, with per-lane dU accumulators when those fit too
// (no cross-lane traffic inside the sweeps).  A shard that fits one workgroup (<= 1024 lanes) runs
// as one workgroup (needed by the persistent trainer, and the cheapest reduction); otherwise
// 4-wave workgroups, shrinking while the LDS image does not fit; if not even one wave fits,
// samples spill to a global [slot][N] workspace (still coalesced along N).
static Geometry choose_geometry(const bsvi_program* p, uint32_t n_local, bool single_block_only) {
    Geometry g;
    const uint32_t waves_total = (n_local + 63) / 64;
    const size_t budget = (size_t)p->max_lds;
    auto fits = [&](uint32_t w, int mode) { return lds_need(p, w, mode) <= budget; };
    const bool one_block = waves_total <= 16;
    if (one_block && fits(waves_total, SM_LACC)) {
        g.n_waves = waves_total; g.n_blocks = 1; g.mode = SM_LACC;
    } else if (single_block_only) {
        if (one_block && fits(waves_total, SM_WSUM)) {
            g.n_waves = waves_total; g.n_blocks = 1; g.mode = SM_WSUM;
        } else if (one_block && fits(waves_total, SM_ZG)) {
            g.n_waves = waves_total; g.n_blocks = 1; g.mode = SM_ZG;
        } else {
            return g;
        }
    } else {
        // several workgroups.  The lane_acc layout (pre-resolved handlers, ~4x cheaper per visit) is worth more
        // than running as one workgroup: 4, 2 or 1 full waves, else a NARROW wave (32 or 16 sample lanes, the
        // rest idle on a shared dummy row — BASELINE config 3, T=200: a sample's row is 3.2 KB); only then the
        // wave_sum layout (one workgroup if the shard fits one), and global slots last.
        uint32_t w = 4;
        while (w > 1 && !fits(w, SM_LACC)) w >>= 1;
        uint32_t narrow = 0;
        if (!fits(1, SM_LACC))
            for (uint32_t l = 32; l >= 16 && !narrow; l >>= 1)
                if (lds_need(p, 1, SM_LACC, l) <= budget) narrow = l;
        if (fits(w, SM_LACC)) {
            g.n_waves = w; g.mode = SM_LACC;
        } else if (narrow) {
            g.n_waves = 1; g.mode = SM_LACC; g.lpw = narrow;
        } else if (one_block && fits(waves_total, SM_WSUM)) {
            g.n_waves = waves_total; g.mode = SM_WSUM;
        } else {
            w = 4;
            while (w > 1 && !fits(w, SM_WSUM)) w >>= 1;
            if (fits(w, SM_WSUM)) {
                g.n_waves = w; g.mode = SM_WSUM;
            } else {
                g.n_waves = 4; g.mode = SM_ZG;
                if (!fits(4, SM_ZG)) return Geometry();
            }
        }
        g.n_blocks = (n_local + g.n_waves * g.lpw - 1) / (g.n_waves * g.lpw);
    }
    g.zglobal = g.mode == SM_ZG;
    // the eps stash rides along whenever the chosen lane_acc geometry still fits with it (generic programs keep
    // regenerating: their kernels do not have the pre-resolved handlers that use it)
    g.stash = g.mode == SM_LACC && !p->generic && lds_need(p, g.n_waves, g.mode, g.lpw, true) <= budget;
    g.lds_bytes = lds_need(p, g.n_waves, g.mode, g.lpw, g.stash);
    g.n_pad = g.n_blocks * g.n_waves * 64;
    return g;
}

static size_t partial_bytes(const bsvi_program* p, const Geometry& g) {
    // rows: one per (wave, share) — covers the launch with one wave per workgroup (share_geometry) — + the row of column sums
    return align_up(((size_t)g.n_blocks * g.n_waves * 8 + 1) * (2 + p->d.n_uniform_grad) * 4, 256);
}

static size_t ws_bytes(const bsvi_program* p, const Geometry& g) {
    if (!g.n_blocks) return 0;
    return partial_bytes(p, g) + (g.zglobal ? 2 * (size_t)p->d.n_slots * g.n_pad * 4 : 0) + 256;
}

static size_t base_ws_bytes(const bsvi_program* p, uint32_t n_local) {
    // cover both the multi-workgroup and the single-workgroup (persistent) geometry
    const size_t a = ws_bytes(p, choose_geometry(p, n_local, false));
    const size_t b = ws_bytes(p, choose_geometry(p, n_local, true));
    return ((a > b ? a : b) + 255) / 256 * 256;
}

// region behind the base workspace used by persistent_multi_kernel (up to 16 workgroups):
// [counter: 256 B][exchange: 2 x 16 x (2 + nUg) floats][parameter copies: 15 x stride][state copies: 15 x 4 x stride]
constexpr uint32_t kMaxMultiWg = 64;     // workgroups of the persistent multi trainer (sample waves x shares)
constexpr uint32_t kMaxMultiWaves = 16;  // sample waves it accepts
static uint32_t mw_stride(const bsvi_program* p) { return (p->d.n_params + 63u) / 64u * 64u + 64u; }
static size_t mw_xchg_floats(const bsvi_program* p) { return 2 * (size_t)kMaxMultiWg * (2 + p->d.n_uniform_grad); }
static size_t mw_bytes(const bsvi_program* p) {
    return 256 + (mw_xchg_floats(p) + (size_t)(kMaxMultiWg - 1) * 5 * mw_stride(p)) * sizeof(float) + 256;
}

extern "C" size_t bsvi_workspace_bytes(const bsvi_program* p, uint32_t n_local) {
    if (!p || !n_local) return 0;
    return base_ws_bytes(p, n_local) + mw_bytes(p) + (p->spec ? bsvi_spec::workspace_bytes(p->spec, n_local) : 0);
}

// the specialised kernels' region of the caller's workspace: behind the interpreter's regions
static void* spec_workspace(const bsvi_program* p, const bsvi_elbo_args* a) {
    if (!a->workspace_dev) return nullptr;
    return (char*)a->workspace_dev + base_ws_bytes(p, a->n_samples_local) + mw_bytes(p);
}

// Try the program-specialised kernel for this call.  Returns 1 when it was launched, 0 when the interpreter should
// run (no specialisation, BSVI_JIT=0, the generated kernel did not compile), or a negative bsvi_status.
static int try_spec(const bsvi_program* p, const bsvi_elbo_args* a, int mode, const bsvi_opt_cfg* cfg, float* params,
                    float* state, const uint8_t* mask, const uint8_t* mask_first, uint32_t pretraining, uint32_t n_iterations,
                    float* loss_slot, float* finite_slot) {
    if (!p->spec || g_debug_stamps || !bsvi_spec::applies(p->spec, a->n_samples_local, mode)) return 0;
    // (caller-weighted gradients — bsvi_elbo_args::f_weight_dev / q_weight_dev, the second pass of a user-defined gradient
    //  estimator — are served by the diagnostic variant of the specialised kernel: one-launch evaluations only)
    if ((a->f_weight_dev || a->q_weight_dev) && mode != bsvi_spec::MODE_SUMS) return 0;
    bsvi_spec::Launch L;
    L.a = a; L.mode = mode; L.cfg = cfg; L.params = params; L.state = state; L.mask = mask; L.mask_first = mask_first;
    L.pretraining_iterations = pretraining; L.n_iterations = n_iterations; L.loss_slot = loss_slot; L.finite_slot = finite_slot;
    L.workspace = spec_workspace(p, a);
    const int rc = bsvi_spec::launch(p->spec, p, L);
    if (rc == BSVI_OK) return 1;
    if (rc == BSVI_ERR_UNSUPPORTED) return 0;
    return rc;
}

// diagnostic hook: device buffer of 10 uint64 receiving (s_memtime, s_memrealtime) at 5 phase
// boundaries of workgroup 0; pass NULL to switch off.  Not part of the product path.
extern "C" void bsvi_debug_set_stamps(unsigned long long* dev) { g_debug_stamps = dev; }

static int fill_kparams(const bsvi_program* p, const bsvi_elbo_args* a, const Geometry& g, KParams& K) {
    if (!a->params_dev && p->d.n_params) return fail(BSVI_ERR_INVALID, "params_dev is null");
    if (!a->obs_dev && p->d.n_obs) return fail(BSVI_ERR_INVALID, "obs_dev is null");
    if (!a->workspace_dev) return fail(BSVI_ERR_INVALID, "workspace_dev is null");
    if (!a->n_samples_local || !a->n_samples_global) return fail(BSVI_ERR_INVALID, "zero samples");
    K.code = p->code; K.aux = p->aux; K.uniform = p->uniform; K.consts = p->consts;
    K.params = a->params_dev; K.obs = a->obs_dev; K.noise = a->noise_dev;
    K.samples_out = a->samples_out_dev; K.noise_out = a->noise_out_dev; K.fvalue_out = a->fvalue_out_dev;
    K.f_weight = a->f_weight_dev; K.q_weight = a->q_weight_dev;
    K.partials = (float*)a->workspace_dev;
    K.zglobal = (float*)((char*)a->workspace_dev + partial_bytes(p, g));
    K.stamps = g_debug_stamps;
    K.fuse_out = nullptr; K.pu_ptr = p->pu_ptr; K.pu_idx = p->pu_idx; K.n_params = p->d.n_params;
    K.n_code = p->d.n_code; K.n_uniform = p->d.n_uniform; K.n_uniform_grad = p->d.n_uniform_grad;
    K.n_slots = p->d.n_slots; K.n_noise = p->d.n_noise; K.n_obs = p->d.n_obs; K.estimator = p->d.estimator;
    K.n_local = a->n_samples_local; K.n_global = a->n_samples_global; K.sample_base = a->sample_base;
    K.n_pad = g.n_pad;
    K.lpw = g.lpw;
    K.stash = g.stash ? 1u : 0u;
    K.seed_lo = (uint32_t)a->seed; K.seed_hi = (uint32_t)(a->seed >> 32);
    K.offset_lo = (uint32_t)a->offset; K.offset_hi = (uint32_t)(a->offset >> 32);
    K.offset_dev = (const unsigned long long*)a->offset_dev;
    K.n_shares = 0;
    for (int v = 0; v < 8; ++v) { K.share_code[v] = nullptr; K.share_aux[v] = nullptr; K.share_n_code[v] = 0; }
    return BSVI_OK;
}

static void fill_rparams(const bsvi_program* p, const Geometry& g, const KParams& K, float* params, float* out,
                         RParams& R) {
    memset(&R, 0, sizeof(R));
    R.uniform = p->uniform; R.pu_ptr = p->pu_ptr; R.pu_idx = p->pu_idx;
    R.partials = K.partials; R.params = params; R.out = out;
    R.n_uniform_grad = p->d.n_uniform_grad; R.n_params = p->d.n_params; R.n_blocks = g.n_blocks;
    R.n_global = K.n_global;
}

static int launch_elbo(const bsvi_program* p, const Geometry& g, const KParams& K_in, hipStream_t s,
                       uint32_t* blocks_out = nullptr) {
    KParams K = K_in;
    const bool out = K.samples_out || K.noise_out || K.fvalue_out || K.stamps;
    uint32_t n_blocks = g.n_blocks;
    // program shares (bsvi_program_set_shares): when every (sample group, share) workgroup still gets a CU of its own,
    // split the model's log-prob records three ways; reduce_kernel adds the extra rows of partial sums like any others
    const char* se = getenv("BSVI_ELBO_SHARES");
    if (p->n_shares > 1 && !K.fuse_out && !out && g.n_blocks * p->n_shares <= 256 && !(se && se[0] == '0')) {
        K.n_shares = p->n_shares;
        for (uint32_t v = 0; v < p->n_shares; ++v) {
            K.share_code[v] = p->shares[v]->code; K.share_aux[v] = p->shares[v]->aux; K.share_n_code[v] = p->shares[v]->d.n_code;
        }
        n_blocks = g.n_blocks * p->n_shares;
    }
    if (blocks_out) *blocks_out = n_blocks;
    dim3 grid(n_blocks), block(g.n_waves * 64);
#define BSVI_LAUNCH_ELBO(SM_, OUT_, GEN_) hipLaunchKernelGGL((elbo_kernel<SM_, OUT_, GEN_>), grid, block, g.lds_bytes, s, K)
#define BSVI_LAUNCH_SM(OUT_, GEN_)                                       \
    do {                                                                 \
        if (g.mode == SM_LACC) BSVI_LAUNCH_ELBO(SM_LACC, OUT_, GEN_);    \
        else if (g.mode == SM_ZG) BSVI_LAUNCH_ELBO(SM_ZG, OUT_, GEN_);   \
        else BSVI_LAUNCH_ELBO(SM_WSUM, OUT_, GEN_);                      \
    } while (0)
    if (p->generic) { if (out) BSVI_LAUNCH_SM(true, true); else BSVI_LAUNCH_SM(false, true); }
    else { if (out) BSVI_LAUNCH_SM(true, false); else BSVI_LAUNCH_SM(false, false); }
#undef BSVI_LAUNCH_SM
#undef BSVI_LAUNCH_ELBO
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}

static int launch_reduce(const bsvi_program* p, const RParams& R, hipStream_t s) {
    const size_t lds = (size_t)p->d.n_uniform_grad * 4 + 16;
    if (lds > (size_t)p->max_lds) return fail(BSVI_ERR_RESOURCE, "uniform-gradient table does not fit LDS");
    if (R.n_blocks > 16) {
        const uint32_t stride = 2 + R.n_uniform_grad;
        float* row = const_cast<float*>(R.partials) + (size_t)R.n_blocks * stride;      // the spare row of the workspace
        hipLaunchKernelGGL(column_sum_kernel, dim3(stride), dim3(256), 0, s, R.partials, row, R.n_blocks, stride);
        RParams R1 = R;
        R1.partials = row;
        R1.n_blocks = 1;
        hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(256), lds, s, R1);
    } else {
        hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(256), lds, s, R);
    }
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}

// With program shares attached, a small shard runs best with ONE wave per workgroup as well (every wave a CU to itself,
// like the multi-workgroup persistent trainer): sample waves x shares workgroups, as long as each still gets a CU.
static Geometry share_geometry(const bsvi_program* p, Geometry g, uint32_t n_local, bool diagnostic) {
    const char* se = getenv("BSVI_ELBO_SHARES");
    if (p->n_shares < 2 || diagnostic || (se && se[0] == '0') || g.mode != SM_LACC || g.lpw != 64 || g.n_waves < 2) return g;
    const uint32_t waves = (n_local + 63) / 64;
    if (waves * p->n_shares > 256) return g;
    Geometry s = g;
    s.n_waves = 1; s.n_blocks = waves;
    s.stash = !p->generic && lds_need(p, 1, SM_LACC, 64, true) <= (size_t)p->max_lds;
    s.lds_bytes = lds_need(p, 1, SM_LACC, 64, s.stash);
    s.n_pad = waves * 64;
    return s;
}

extern "C" int bsvi_elbo_fwd_bwd(const bsvi_program* p, const bsvi_elbo_args* a) {
    if (a) BSVI_CHECK_STRUCT(a, bsvi_elbo_args);
    if (!p || !a) return fail(BSVI_ERR_INVALID, "null argument");
    if (!a->out_dev) return fail(BSVI_ERR_INVALID, "out_dev is null");
    if (a->q_weight_dev && p->d.estimator != BSVI_EST_BLACKBOX)
        return fail(BSVI_ERR_INVALID, "q_weight_dev needs a program lowered for the BlackBox estimator (its records accumulate log q)");
    {
        const int sp = try_spec(p, a, bsvi_spec::MODE_SUMS, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr, nullptr);
        if (sp) return sp < 0 ? sp : BSVI_OK;
    }
    Geometry g = choose_geometry(p, a->n_samples_local, false);
    if (!g.n_blocks) return fail(BSVI_ERR_RESOURCE, "program does not fit the LDS budget");
    const char* se = getenv("BSVI_ELBO_SHARES");
    const bool diag = a->samples_out_dev || a->noise_out_dev || a->fvalue_out_dev || g_debug_stamps;
    g = share_geometry(p, g, a->n_samples_local, diag);
    KParams K;
    int rc = fill_kparams(p, a, g, K);
    if (rc) return rc;
    const bool shares = p->n_shares > 1 && !diag && !(se && se[0] == '0');
    if (g.n_blocks == 1 && !shares) {
        K.fuse_out = a->out_dev;          // one workgroup: its epilogue writes the output block, one launch in all
        return launch_elbo(p, g, K, (hipStream_t)a->stream);
    }
    // (with program shares the three workgroups + reduce_kernel beat the single fused workgroup: 33.9 vs 43.5 us at cfg 1)
    uint32_t rows = g.n_blocks;
    rc = launch_elbo(p, g, K, (hipStream_t)a->stream, &rows);
    if (rc) return rc;
    RParams R;
    fill_rparams(p, g, K, (float*)a->params_dev, a->out_dev, R);
    R.n_blocks = rows;
    return launch_reduce(p, R, (hipStream_t)a->stream);
}

extern "C" int bsvi_finalize(const bsvi_program* p, float* out_dev, uint32_t n_global, void* stream) {
    if (!p || !out_dev || !n_global) return fail(BSVI_ERR_INVALID, "null argument");
    const uint32_t n = p->d.n_params ? p->d.n_params : 1;
    hipLaunchKernelGGL(finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, out_dev,
                       p->d.n_params, n_global);
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}

static int check_cfg(const bsvi_opt_cfg* cfg) {
    if (!cfg) return fail(BSVI_ERR_INVALID, "null optimizer config");
    if (cfg->kind > BSVI_OPT_ADAM) return fail(BSVI_ERR_UNSUPPORTED, "unknown optimizer kind");
    return BSVI_OK;
}

extern "C" int bsvi_optimizer_step(const bsvi_opt_cfg* cfg, float* params_dev, const float* out_dev, float* state_dev,
                                   const uint8_t* active_mask_dev, uint32_t n_params, void* stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!n_params) return BSVI_OK;
    if (!params_dev || !out_dev || !state_dev || !active_mask_dev) return fail(BSVI_ERR_INVALID, "null argument");
    hipLaunchKernelGGL(optimizer_kernel, dim3((n_params + 255) / 256), dim3(256), 0, (hipStream_t)stream, *cfg,
                       params_dev, out_dev, state_dev, active_mask_dev, n_params);
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}

extern "C" int bsvi_finalize_step(const bsvi_opt_cfg* cfg, float* params_dev, float* out_dev, float* state_dev,
                                  const uint8_t* active_mask_dev, uint32_t n_params, uint32_t n_samples_global,
                                  float* loss_slot_dev, float* finite_slot_dev, void* stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!out_dev || !n_samples_global) return fail(BSVI_ERR_INVALID, "null argument");
    if (n_params && (!params_dev || !state_dev || !active_mask_dev)) return fail(BSVI_ERR_INVALID, "null argument");
    const uint32_t n = n_params ? n_params : 1;
    hipLaunchKernelGGL(finalize_step_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *cfg, params_dev,
                       out_dev, state_dev, active_mask_dev, n_params, n_samples_global, loss_slot_dev, finite_slot_dev);
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}

extern "C" int bsvi_finalize_step_counted(const bsvi_opt_cfg* cfg, float* params_dev, float* out_dev, float* state_dev,
                                          const uint8_t* active_mask_dev, const uint8_t* active_mask_first_dev,
                                          uint32_t pretraining_iterations, uint32_t n_params, uint32_t n_samples_global,
                                          float* loss_curve_dev, float* finite_curve_dev, uint64_t* counters_dev,
                                          void* stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!out_dev || !n_samples_global || !counters_dev) return fail(BSVI_ERR_INVALID, "null argument");
    if (n_params && (!params_dev || !state_dev || !active_mask_dev || !active_mask_first_dev))
        return fail(BSVI_ERR_INVALID, "null argument");
    if (n_params > 1024) return fail(BSVI_ERR_UNSUPPORTED, "bsvi_finalize_step_counted: at most 1024 parameters (one workgroup)");
    const uint32_t threads = n_params <= 64 ? 64 : (n_params + 63) / 64 * 64;
    hipLaunchKernelGGL(finalize_step_counted_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, *cfg, params_dev, out_dev,
                       state_dev, active_mask_dev, active_mask_first_dev, pretraining_iterations, n_params, n_samples_global,
                       loss_curve_dev, finite_curve_dev, (unsigned long long*)counters_dev);
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}

extern "C" int bsvi_svi_step(const bsvi_program* p, const bsvi_elbo_args* a, const bsvi_opt_cfg* cfg,
                             float* params_dev, float* state_dev, const uint8_t* active_mask_dev,
                             float* loss_slot_dev, float* finite_slot_dev) {
    if (a) BSVI_CHECK_STRUCT(a, bsvi_elbo_args);
    if (!p || !a) return fail(BSVI_ERR_INVALID, "null argument");
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!a->out_dev || !params_dev || !state_dev || !active_mask_dev) return fail(BSVI_ERR_INVALID, "null argument");
    if (a->n_samples_local != a->n_samples_global) return fail(BSVI_ERR_INVALID, "bsvi_svi_step is the single-GPU path");
    {
        const int sp = try_spec(p, a, bsvi_spec::MODE_STEP, cfg, params_dev, state_dev, active_mask_dev, active_mask_dev, 0, 1,
                                loss_slot_dev, finite_slot_dev);
        if (sp) return sp < 0 ? sp : BSVI_OK;
    }
    Geometry g = choose_geometry(p, a->n_samples_local, false);
    if (!g.n_blocks) return fail(BSVI_ERR_RESOURCE, "program does not fit the LDS budget");
    g = share_geometry(p, g, a->n_samples_local, a->samples_out_dev || a->noise_out_dev || a->fvalue_out_dev || g_debug_stamps);
    KParams K;
    bsvi_elbo_args aa = *a;
    aa.params_dev = params_dev;
    rc = fill_kparams(p, &aa, g, K);
    if (rc) return rc;
    uint32_t rows = g.n_blocks;
    rc = launch_elbo(p, g, K, (hipStream_t)a->stream, &rows);
    if (rc) return rc;
    RParams R;
    fill_rparams(p, g, K, params_dev, a->out_dev, R);
    R.n_blocks = rows;
    R.state = state_dev; R.active_mask = active_mask_dev; R.loss_slot = loss_slot_dev; R.finite_slot = finite_slot_dev;
    R.do_finalize = 1; R.do_step = 1; R.cfg = *cfg;
    return launch_reduce(p, R, (hipStream_t)a->stream);
}

// persistent_multi_kernel applies: 5..16 waves of 64 sample lanes, one lane-accumulator wave fits a CU's LDS
static bool multi_persistent_applies(const bsvi_program* p, uint32_t n_local) {
    const char* e = getenv("BSVI_PERSISTENT_MULTI");       // read per call: tests and tools flip it
    if (e && e[0] == '0') return false;
    const uint32_t waves = (n_local + 63) / 64;
    if (!(waves >= 5 && waves <= kMaxMultiWaves && lds_need(p, 1, SM_LACC, 64, false) <= (size_t)p->max_lds)) return false;
    // the in-kernel exchange needs every workgroup resident: each takes most of a CU's LDS, so one CU apiece
    static int n_cus = 0;
    if (!n_cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cus = 64;
    }
    return waves * 3u <= (uint32_t)n_cus;        // (up to three program shares per sample wave)
}

extern "C" int bsvi_persistent_supported(const bsvi_program* p, uint32_t n_local) {
    if (!p || !n_local) return 0;
    if (p->spec && bsvi_spec::applies(p->spec, n_local, bsvi_spec::MODE_LOOP)) return 1;
    if (multi_persistent_applies(p, n_local)) return 1;
    // One launch for the whole loop pays when the single workgroup runs the fast layout, or when several
    // workgroups would not run it either; a spilled (global-slot) workgroup is always slower than many LDS ones.
    const Geometry g = choose_geometry(p, n_local, true), m = choose_geometry(p, n_local, false);
    if (g.n_blocks != 1 || g.mode == SM_ZG) return 0;
    return (g.mode == SM_LACC || m.mode != SM_LACC) ? 1 : 0;
}

extern "C" int bsvi_train_persistent(const bsvi_program* p, const bsvi_elbo_args* a, const bsvi_opt_cfg* cfg,
                                     float* params_dev, float* state_dev, const uint8_t* active_mask_dev,
                                     uint32_t n_iterations, float* loss_curve_dev, float* finite_dev) {
    if (a) BSVI_CHECK_STRUCT(a, bsvi_elbo_args);
    return bsvi_train_persistent2(p, a, cfg, params_dev, state_dev, active_mask_dev, active_mask_dev, 0,
                                  n_iterations, loss_curve_dev, finite_dev);
}

static int train_persistent_impl(const bsvi_program* p, const bsvi_program* const* shares, uint32_t n_shares,
                                 const bsvi_elbo_args* a, const bsvi_opt_cfg* cfg,
                                 float* params_dev, float* state_dev, const uint8_t* active_mask_dev,
                                 const uint8_t* active_mask_first_dev, uint32_t pretraining_iterations,
                                 uint32_t n_iterations, float* loss_curve_dev, float* finite_dev);

extern "C" int bsvi_train_persistent2(const bsvi_program* p, const bsvi_elbo_args* a, const bsvi_opt_cfg* cfg,
                                      float* params_dev, float* state_dev, const uint8_t* active_mask_dev,
                                      const uint8_t* active_mask_first_dev, uint32_t pretraining_iterations,
                                      uint32_t n_iterations, float* loss_curve_dev, float* finite_dev) {
    if (a) BSVI_CHECK_STRUCT(a, bsvi_elbo_args);
    return train_persistent_impl(p, nullptr, 0, a, cfg, params_dev, state_dev, active_mask_dev, active_mask_first_dev,
                                 pretraining_iterations, n_iterations, loss_curve_dev, finite_dev);
}

// K iterations of the SHARDED loop in one launch per rank: this rank's samples, the cross-rank sum of the loss words and the
// per-parameter gradient sums INSIDE the loop (spec_main.h, spec_exchange: the one-shot direct-write exchange run by the
// owners' wave), the replicated optimizer step.  Served by the program-specialised one-workgroup kernel only.
extern "C" int bsvi_train_persistent_exchange(const bsvi_program* p, const bsvi_elbo_args* a, const bsvi_opt_cfg* cfg,
                                              float* params_dev, float* state_dev, const uint8_t* active_mask_dev,
                                              const uint8_t* active_mask_first_dev, uint32_t pretraining_iterations,
                                              uint32_t n_iterations, float* loss_curve_dev, float* finite_dev, bsvi_exchange* x) {
    if (!p || !a || !x) return fail(BSVI_ERR_INVALID, "null argument");
    BSVI_CHECK_STRUCT(a, bsvi_elbo_args);
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!a->out_dev || !params_dev || !active_mask_dev || !active_mask_first_dev || !loss_curve_dev || !finite_dev)
        return fail(BSVI_ERR_INVALID, "null argument");
    if (a->offset_dev) return fail(BSVI_ERR_INVALID, "the in-kernel training loop counts its own iterations: offset_dev must be null");
    if (a->noise_dev || a->samples_out_dev || a->noise_out_dev || a->fvalue_out_dev || a->f_weight_dev || a->q_weight_dev)
        return fail(BSVI_ERR_UNSUPPORTED, "the in-loop exchange serves Philox noise without per-sample outputs");
    if (!p->spec || g_debug_stamps || !bsvi_spec::applies(p->spec, a->n_samples_local, bsvi_spec::MODE_LOOP))
        return fail(BSVI_ERR_UNSUPPORTED, "this shard does not run on the program-specialised one-workgroup kernel");
    const void* desc = bsvi_exchange_descriptor(x, BSVI_OUT_HEADER + p->d.n_params);
    if (!desc) return BSVI_ERR_INVALID;
    bsvi_spec::Launch L;
    L.a = a; L.mode = bsvi_spec::MODE_LOOP; L.cfg = cfg; L.params = params_dev; L.state = state_dev; L.mask = active_mask_dev;
    L.mask_first = active_mask_first_dev; L.pretraining_iterations = pretraining_iterations; L.n_iterations = n_iterations;
    L.loss_slot = loss_curve_dev; L.finite_slot = finite_dev; L.workspace = spec_workspace(p, a); L.xchg = desc;
    return bsvi_spec::launch(p->spec, p, L);
}

// Attach the shares of a program (programs created from lowering.Program.shares[n]: identical tables, own code) for the
// multi-workgroup launches of bsvi_elbo_fwd_bwd / bsvi_svi_step.  The shares must outlive `p`'s use.
extern "C" int bsvi_program_set_shares(bsvi_program* p, const bsvi_program* const* shares, uint32_t n_shares) {
    if (!p) return fail(BSVI_ERR_INVALID, "null argument");
    if (!shares || n_shares < 2) { p->n_shares = 0; return BSVI_OK; }
    if (n_shares > 8) return fail(BSVI_ERR_INVALID, "at most 8 program shares");
    for (uint32_t v = 0; v < n_shares; ++v) {
        if (!shares[v]) return fail(BSVI_ERR_INVALID, "null program share");
        if (shares[v]->d.n_uniform != p->d.n_uniform || shares[v]->d.n_uniform_grad != p->d.n_uniform_grad ||
            shares[v]->d.n_slots != p->d.n_slots || shares[v]->d.n_params != p->d.n_params ||
            shares[v]->d.n_code > p->d.n_code || (shares[v]->generic && !p->generic))
            return fail(BSVI_ERR_INVALID, "a program share does not match the program's tables");
    }
    for (uint32_t v = 0; v < n_shares; ++v) p->shares[v] = shares[v];
    p->n_shares = n_shares;
    return BSVI_OK;
}

// how many shares of the program (2, 3 or 1 = none) the multi-workgroup trainer would run for this sample count
extern "C" int bsvi_persistent_split_shares(const bsvi_program* p, uint32_t n_local) {
    if (!p || !n_local || !multi_persistent_applies(p, n_local)) return 1;
    if (p->spec && bsvi_spec::applies(p->spec, n_local, bsvi_spec::MODE_LOOP)) return 1;
    const char* e = getenv("BSVI_PERSISTENT_SHARES");
    const uint32_t waves = (n_local + 63) / 64;
    // more shares mean more workgroups at the in-kernel exchange, which every workgroup walks in full: 2-3 shares
    // are the optimum there (32.3 us at cfg 1; 34.9 with 6, 37.8 with 8) — the launch-per-iteration path, whose
    // reduce_kernel adds the rows once, is the one that profits from 8 (29.0 us)
    uint32_t want = e ? (uint32_t)atoi(e) : 2u;
    if (want > 8) want = 8;
    while (want > 1 && (waves * want > kMaxMultiWg || want == 5 || want == 7)) --want;      // the lowering emits 2, 3, 4, 6, 8
    return (int)(want < 1 ? 1 : want);
}

// the multi-workgroup trainer with the model's log-prob records split over `n_shares` programs (same tables, every
// share samples the posterior; lowering.Program.shares): shares[0..n_shares) are programs created from the shares
extern "C" int bsvi_train_persistent_split(const bsvi_program* p, const bsvi_program* const* shares, uint32_t n_shares,
                                           const bsvi_elbo_args* a, const bsvi_opt_cfg* cfg,
                                           float* params_dev, float* state_dev, const uint8_t* active_mask_dev,
                                           const uint8_t* active_mask_first_dev, uint32_t pretraining_iterations,
                                           uint32_t n_iterations, float* loss_curve_dev, float* finite_dev) {
    if (a) BSVI_CHECK_STRUCT(a, bsvi_elbo_args);
    if (!shares || n_shares < 2 || n_shares > 8) return fail(BSVI_ERR_INVALID, "2..8 program shares expected");
    for (uint32_t v = 0; v < n_shares; ++v) {
        if (!shares[v]) return fail(BSVI_ERR_INVALID, "null program share");
        if (shares[v]->d.n_uniform != p->d.n_uniform || shares[v]->d.n_uniform_grad != p->d.n_uniform_grad ||
            shares[v]->d.n_slots != p->d.n_slots || shares[v]->d.n_params != p->d.n_params ||
            shares[v]->d.n_code > p->d.n_code || (shares[v]->generic && !p->generic))
            return fail(BSVI_ERR_INVALID, "a program share does not match the program's tables");
    }
    return train_persistent_impl(p, shares, n_shares, a, cfg, params_dev, state_dev, active_mask_dev, active_mask_first_dev,
                                 pretraining_iterations, n_iterations, loss_curve_dev, finite_dev);
}

static int train_persistent_impl(const bsvi_program* p, const bsvi_program* const* shares, uint32_t n_shares,
                                 const bsvi_elbo_args* a, const bsvi_opt_cfg* cfg,
                                 float* params_dev, float* state_dev, const uint8_t* active_mask_dev,
                                 const uint8_t* active_mask_first_dev, uint32_t pretraining_iterations,
                                 uint32_t n_iterations, float* loss_curve_dev, float* finite_dev) {
    if (!p || !a) return fail(BSVI_ERR_INVALID, "null argument");
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!a->out_dev || !params_dev || !active_mask_dev || !active_mask_first_dev || !loss_curve_dev || !finite_dev)
        return fail(BSVI_ERR_INVALID, "null argument");
    if (a->n_samples_local != a->n_samples_global) return fail(BSVI_ERR_INVALID, "persistent trainer is the single-GPU path");
    if (a->offset_dev) return fail(BSVI_ERR_INVALID, "the in-kernel training loop counts its own iterations: offset_dev must be null");
    {
        const int sp = try_spec(p, a, bsvi_spec::MODE_LOOP, cfg, params_dev, state_dev, active_mask_dev, active_mask_first_dev,
                                pretraining_iterations, n_iterations, loss_curve_dev, finite_dev);
        if (sp) return sp < 0 ? sp : BSVI_OK;
    }
    // state_dev == NULL (a fresh optimizer whose final state the caller does not want) is served by the program-specialised
    // kernel only (bsvi_program_engine tells which engine a call would run on)
    if (!state_dev) return fail(BSVI_ERR_INVALID, "state_dev is null and this call does not run on the specialised kernel");
    const bool multi = multi_persistent_applies(p, a->n_samples_local);
    Geometry g = choose_geometry(p, a->n_samples_local, true);
    if (multi && g.n_blocks != 1) {      // does not fit ONE workgroup, but one wave per workgroup does
        g = Geometry();
        g.n_waves = (a->n_samples_local + 63) / 64; g.n_blocks = 1; g.mode = SM_LACC; g.lpw = 64;
    }
    if (g.n_blocks != 1) return fail(BSVI_ERR_RESOURCE, "sample count does not fit one workgroup");
    PParams P;
    memset(&P, 0, sizeof(P));
    bsvi_elbo_args aa = *a;
    aa.params_dev = params_dev;
    rc = fill_kparams(p, &aa, g, P.K);
    if (rc) return rc;
    fill_rparams(p, g, P.K, params_dev, a->out_dev, P.R);
    P.R.state = state_dev; P.R.active_mask = active_mask_dev; P.R.cfg = *cfg; P.R.do_finalize = 1; P.R.do_step = 1;
    P.n_iterations = n_iterations; P.loss_curve = loss_curve_dev; P.finite_curve = finite_dev;
    P.active_mask_first = active_mask_first_dev; P.pretraining_iterations = pretraining_iterations;
    hipStream_t st = (hipStream_t)a->stream;
    // five or more waves in ONE workgroup put two of them on one SIMD of the CU, and that SIMD sets the pace: give
    // every wave its own CU instead (persistent_multi_kernel; BSVI_PERSISTENT_MULTI=0 keeps the single workgroup)
    if (shares && !multi) return fail(BSVI_ERR_INVALID, "program shares need the multi-workgroup trainer");
    if (multi) {
        const uint32_t sample_waves = g.n_waves;
        const uint32_t n_wg = sample_waves * (shares ? n_shares : 1u);
        if (n_wg > kMaxMultiWg) return fail(BSVI_ERR_INVALID, "too many workgroups for the exchange region");
        if (shares) {
            P.n_shares = n_shares;
            for (uint32_t v = 0; v < n_shares; ++v) {
                P.share_code[v] = shares[v]->code; P.share_aux[v] = shares[v]->aux; P.share_n_code[v] = shares[v]->d.n_code;
            }
        }
        Geometry g1 = g;
        g1.n_waves = 1; g1.n_blocks = n_wg;
        g1.stash = !p->generic && lds_need(p, 1, SM_LACC, 64, true) <= (size_t)p->max_lds;
        g1.lds_bytes = lds_need(p, 1, SM_LACC, 64, g1.stash);
        g1.n_pad = n_wg * 64;
        rc = fill_kparams(p, &aa, g1, P.K);
        if (rc) return rc;
        char* region = (char*)a->workspace_dev + base_ws_bytes(p, a->n_samples_local);
        P.mw_counter = (uint32_t*)region;
        P.mw_xchg = (float*)(region + 256);
        P.mw_params = P.mw_xchg + mw_xchg_floats(p);
        P.mw_stride = mw_stride(p);
        P.mw_state = P.mw_params + (size_t)(kMaxMultiWg - 1) * P.mw_stride;
        HIP_TRY(hipMemsetAsync(region, 0, 256 + mw_xchg_floats(p) * sizeof(float), st));
        dim3 mgrid(n_wg), mblock(64);
        if (p->generic) hipLaunchKernelGGL((persistent_multi_kernel<SM_LACC, true>), mgrid, mblock, g1.lds_bytes, st, P);
        else hipLaunchKernelGGL((persistent_multi_kernel<SM_LACC, false>), mgrid, mblock, g1.lds_bytes, st, P);
        HIP_TRY(hipGetLastError());
        return BSVI_OK;
    }
    dim3 grid(1), block(g.n_waves * 64);
#define BSVI_LAUNCH_P(SM_, GEN_) hipLaunchKernelGGL((persistent_kernel<SM_, GEN_>), grid, block, g.lds_bytes, st, P)
    if (p->generic) {
        if (g.mode == SM_LACC) BSVI_LAUNCH_P(SM_LACC, true); else if (g.mode == SM_ZG) BSVI_LAUNCH_P(SM_ZG, true); else BSVI_LAUNCH_P(SM_WSUM, true);
    } else {
        if (g.mode == SM_LACC) BSVI_LAUNCH_P(SM_LACC, false); else if (g.mode == SM_ZG) BSVI_LAUNCH_P(SM_ZG, false); else BSVI_LAUNCH_P(SM_WSUM, false);
    }
#undef BSVI_LAUNCH_P
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}

// ---- program specialisation: introspection (tests, bench) -------------------------------------------
extern "C" size_t bsvi_program_source(const bsvi_program_desc* desc, int variant, char* buf, size_t capacity) {
    if (validate(desc) != BSVI_OK) return 0;
    std::string why;
    bsvi_spec::Spec* sp = bsvi_spec::create(*desc, why);
    if (!sp) { fail(BSVI_ERR_UNSUPPORTED, "program not specialised: " + why); return 0; }
    const std::string& src = bsvi_spec::source(sp, variant);
    const size_t need = src.size() + 1;
    if (buf && capacity >= need) memcpy(buf, src.c_str(), need);
    bsvi_spec::destroy(sp);
    return need;
}

extern "C" int bsvi_jit_compile(const char* source, size_t* code_bytes) {
    if (!source) return fail(BSVI_ERR_INVALID, "null source");
    std::vector<char> code;
    std::string log;
    const int rc = bsvi_spec::compile(source, code, log);
    if (rc) return fail(rc, log);
    if (code_bytes) *code_bytes = code.size();
    return BSVI_OK;
}

extern "C" int bsvi_jit_load(const char* source, size_t* code_bytes, int* origin) {
    if (!source) return fail(BSVI_ERR_INVALID, "null source");
    std::vector<char> code;
    std::string log;
    const int rc = bsvi_spec::obtain(source, code, log, origin);
    if (rc) return fail(rc, log);
    if (code_bytes) *code_bytes = code.size();
    return BSVI_OK;
}

extern "C" int bsvi_jit_last_origin(void) { return bsvi_spec::last_origin(); }

extern "C" size_t bsvi_jit_cache_dir(char* buf, size_t capacity) {
    const std::string dir = bsvi_spec::cache_directory();
    if (buf && capacity > dir.size()) memcpy(buf, dir.c_str(), dir.size() + 1);
    return dir.size() + 1;
}

extern "C" size_t bsvi_jit_compiler_identity(char* buf, size_t capacity) {
    const std::string id = bsvi_spec::compiler_identity();
    if (buf && capacity > id.size()) memcpy(buf, id.c_str(), id.size() + 1);
    return id.size() + 1;
}

extern "C" int bsvi_program_engine(const bsvi_program* p, uint32_t n_local, int mode, uint32_t* n_blocks, uint32_t* n_threads,
                                   uint32_t* lds_bytes) {
    if (!p) return 0;
    if (!p->spec || !bsvi_spec::applies(p->spec, n_local, mode)) return 0;
    bsvi_spec::geometry(p->spec, n_local, mode, n_blocks, n_threads, lds_bytes);
    return 1;
}

extern "C" int bsvi_max_lds_bytes(const bsvi_program* p) { return p ? p->max_lds : 0; }

// launch geometry the library would use — exported for tests, bench and DESIGN.md tables
extern "C" int bsvi_query_geometry(const bsvi_program* p, uint32_t n_local, uint32_t* n_blocks, uint32_t* n_waves,
                                   uint32_t* zglobal, uint64_t* lds_bytes) {
    if (!p) return fail(BSVI_ERR_INVALID, "null argument");
    Geometry g = choose_geometry(p, n_local, false);
    if (!g.n_blocks) return fail(BSVI_ERR_RESOURCE, "program does not fit the LDS budget");
    if (n_blocks) *n_blocks = g.n_blocks;
    if (n_waves) *n_waves = g.n_waves;
    // low byte: 0 LDS rows + wave sums, 1 LDS rows + lane accumulators, 2 global slots; next byte: sample lanes per wave
    if (zglobal) *zglobal = (uint32_t)g.mode | (g.lpw << 8);
    if (lds_bytes) *lds_bytes = g.lds_bytes;
    return BSVI_OK;
}

// test hook (not part of the product path): out is [4][n] = value, d/dx, d/dp0, d/dp1
extern "C" int bsvi_debug_math(int fn, int dist, const float* x_dev, const float* p0_dev, const float* p1_dev,
                               float* out_dev, uint32_t n, void* stream) {
    if (!x_dev || !p0_dev || !p1_dev || !out_dev || !n) return fail(BSVI_ERR_INVALID, "null argument");
    hipLaunchKernelGGL(debug_math_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, fn, dist, x_dev,
                       p0_dev, p1_dev, out_dev, n);
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}

#include "dense_kernel.inc"
#include "bnn_kernel.inc"

// ---- the minibatch data path of the scalar engine (standard_variables.py:71-112, distributions.py:393-473) ------------
// A variable observed through an EmpiricalVariable sees `batch` rows of a dataset, other rows in every evaluation.  To the per-sample
// program they are ordinary observations; this launch refreshes that stretch of the observation buffer in front of the evaluation:
// row i of the minibatch is dataset row `minibatch_index(i)` — the keyed bijection of [0, DS) the dense path's dense_head draws with
// the same (seed, offset), sampling without replacement like np.random.choice(replace=False) — or the caller's.
namespace bsvi {
__global__ void __launch_bounds__(256) minibatch_gather_kernel(const float* dataset, uint32_t DS, uint32_t row, uint32_t B, const int32_t* indices_in,
                                                                uint32_t seed_lo, uint32_t seed_hi, uint32_t off_lo, uint32_t off_hi, float* dst, int32_t* indices_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= B * row) return;
    const uint32_t b = i / row, e = i - b * row;
    DParams Dm;
    Dm.indices_in = indices_in; Dm.DS = DS;
    Dm.offset_lo = off_lo; Dm.offset_hi = off_hi; Dm.seed_lo = seed_lo; Dm.seed_hi = seed_hi;
    const uint32_t r = minibatch_index(Dm, b);
    dst[i] = dataset[(size_t)r * row + e];
    if (indices_out && e == 0u) indices_out[b] = (int32_t)r;
}
}  // namespace bsvi

extern "C" int bsvi_minibatch_gather(const float* dataset_dev, uint32_t dataset_size, uint32_t row_floats, uint32_t batch_size,
                                     const int32_t* indices_dev, uint64_t seed, uint64_t offset, float* dst_dev, int32_t* indices_out_dev,
                                     void* stream) {
    if (!dataset_dev || !dst_dev || !dataset_size || !row_floats || !batch_size) return fail(BSVI_ERR_INVALID, "null argument");
    if (batch_size > dataset_size) return fail(BSVI_ERR_INVALID, "bsvi_minibatch_gather: batch_size exceeds dataset_size");
    if ((uint64_t)batch_size * row_floats >= (1ull << 31)) return fail(BSVI_ERR_RESOURCE, "bsvi_minibatch_gather: minibatch too large");
    const uint32_t n = batch_size * row_floats;
    hipLaunchKernelGGL(bsvi::minibatch_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dataset_dev, dataset_size,
                       row_floats, batch_size, indices_dev, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)offset, (uint32_t)(offset >> 32),
                       dst_dev, indices_out_dev);
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}
