// spec_main.h — second half of the frame around a program-specialised ELBO kernel (see spec_prelude.h): the kernel.
//
// One workgroup = up to SPEC_MAX_WAVES waves of 64 Monte-Carlo samples.  Per iteration:
//   prologue   U[k] = a + b*g(theta | const), 1/U[k], log U[k]     (geometric_ranges.py transforms: once per launch for
//              every entry, then per iteration by the thread that owns the parameter)
//   body       spec_draw() + spec_body(): GENERATED straight-line forward / reverse sweep of one sample per lane,
//              registers only; per-lane gradient contributions leave through the transpose tile (SPEC_DU)
//   reduce     wave (DPP) -> workgroup (LDS) -> grid (partials + arrival ticket, last workgroup finishes); fixed
//              order everywhere, no float atomics: bitwise reproducible
//   epilogue   chain rule U -> theta through the CSR map; by mode: sums only (the caller all-reduces them over the
//              sample shards of other GPUs), or loss / finite flag / scaled gradients / fused SGD-Adam step
//              (inference.py:96-108), or that whole loop n_iterations times inside this one launch.
#pragma once

namespace bsvi {

// Scalar-register discipline (as in the interpreter's persistent kernels): SpecArgs is ~50 dwords.  Read as `A.x`
// they are all loaded at entry and stay live through the body, whose Philox key schedule and exec masks then spill
// through v_writelane / v_readlane.  So every phase re-reads what it needs from the kernarg segment, through a
// pointer the optimiser cannot see through.
#if defined(__HIP_DEVICE_COMPILE__)
#define SPEC_CONST_AS __attribute__((address_space(4)))
#define SPEC_RELOAD_ARGS() asm volatile("" : "+s"(ka))
#else
#define SPEC_CONST_AS
#define SPEC_RELOAD_ARGS()
#endif

// TAB (words): uniform entries [4 * NU] | CSR row pointers [NP + 1] | positions [NUG] | uniform indices [NUG] |
//              per parameter: bit 0 active, bit 1 active while iteration <= pretraining_iterations [NP]
#define SPEC_TAB_PTR (4 * SPEC_N_UNIFORM)
#define SPEC_TAB_POS (SPEC_TAB_PTR + SPEC_N_PARAMS + 1)
#define SPEC_TAB_IDX (SPEC_TAB_POS + SPEC_N_POS)
#define SPEC_TAB_MASK (SPEC_TAB_IDX + SPEC_N_POS)

// The barriers of the iteration loop order LDS traffic only (sums, uniform table, theta all live in LDS), so they wait
// for the LDS counter and not for global memory: __syncthreads() also drains vmcnt, i.e. every iteration would wait
// for the write acknowledgements of its loss / gradient stores — 1.3 of 6.3 us per iteration at BASELINE config 1.
__device__ __forceinline__ void spec_lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// U[k] = a + b * g(x) and its companions, from the LDS copy of entry k
// The transforms of the uniform table: values through the library forms of dist_math.h (softplusf_ / sigmoidf_, see
// there), derivatives on the hardware transcendental units (their rounding moves a gradient by 1e-7).  The transform id is
// data: written as one switch the compiler evaluates EVERY case and selects — tanh polynomial, sqrt refinement and all;
// the rare ones stay out of line behind a real branch.
__device__ __noinline__ float spec_utransform_rare(uint32_t t, float x) { return utransform((int)t, x); }
__device__ __noinline__ float spec_utransform_grad_rare(uint32_t t, float x) { return utransform_grad((int)t, x); }
#ifndef SPEC_RARE_TRANSFORMS
#define SPEC_RARE_TRANSFORMS 1
#endif
#ifndef SPEC_UT_MASK
#define SPEC_UT_MASK 0xFFu
#endif
__device__ __forceinline__ float utransform_common(uint32_t t, float x) {
#if SPEC_RARE_TRANSFORMS
    if (t > BSVI_UT_SIGMOID) return spec_utransform_rare(t, x);
#endif
    // (both evaluated and selected — unless the program's table has no entry of the kind: SPEC_UT_MASK)
    const float soft = ((SPEC_UT_MASK >> BSVI_UT_SOFTPLUS) & 1u) ? softplusf_(x) : x;
    const float sig = ((SPEC_UT_MASK >> BSVI_UT_SIGMOID) & 1u) ? sigmoidf_(x) : x;
    return t == BSVI_UT_IDENTITY ? x : (t == BSVI_UT_SOFTPLUS ? soft : sig);
}
__device__ __forceinline__ float spec_utransform_grad(uint32_t t, float x) {
#if SPEC_RARE_TRANSFORMS
    if (t > BSVI_UT_SIGMOID) return spec_utransform_grad_rare(t, x);
#endif
    const float s = spec_rcp(1.0f + __expf(-x));               // sigmoid(x) = softplus'(x)
    const float soft = x > 20.0f ? 1.0f : s;
    return t == BSVI_UT_IDENTITY ? 1.0f : (t == BSVI_UT_SOFTPLUS ? soft : s * (1.0f - s));
}
__device__ __forceinline__ void spec_store_uniform(uint32_t k, float u) {
    spec_lds[k] = u;
    spec_lds[SPEC_OFF_UR + k] = spec_rcp(u);
    spec_lds[SPEC_OFF_UL + k] = spec_log(u);
}
__device__ __forceinline__ void spec_publish_uniform(const uint32_t* TAB, uint32_t k, float x) {
    const uint32_t w1 = TAB[4 * k + 1];
    spec_store_uniform(k, __uint_as_float(TAB[4 * k + 2]) + __uint_as_float(TAB[4 * k + 3]) * utransform_common(w1 & 0xFFu, x));
}

// What the thread that owns parameter `tid` keeps in registers across the iterations of a launch (parameters beyond
// the workgroup size, or with more than two uniform entries, go through the LDS working copy instead)
struct SpecOwn {
    float theta, s0, s1, s2, st;     // parameter and optimizer state
    float a[2], b[2];                // its uniform entries: U = a + b * g(theta)
    uint32_t pos[2], k[2], tr[2];
    uint32_t n, mask;                // entries (0..2); mask bits (SPEC_TAB_MASK)
    double p1, p2;                   // Adam: beta1^st, beta2^st as running products (optimizer_apply_running)
};
__device__ __forceinline__ void spec_own_store_products(spec_f4* row, const SpecOwn& o) {
    const unsigned long long a = __double_as_longlong(o.p1), b = __double_as_longlong(o.p2);
    row[4] = spec_f4{__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)b), __uint_as_float((uint32_t)(b >> 32))};
}
__device__ __forceinline__ void spec_own_store(spec_f4* row, const SpecOwn& o) {
    row[0] = spec_f4{o.theta, o.s0, o.s1, o.s2};
    row[1] = spec_f4{o.st, o.a[0], o.b[0], o.a[1]};
    row[2] = spec_f4{o.b[1], __uint_as_float(o.pos[0]), __uint_as_float(o.pos[1]), __uint_as_float(o.k[0])};
    row[3] = spec_f4{__uint_as_float(o.k[1]), __uint_as_float(o.tr[0] | (o.tr[1] << 8) | (o.n << 16) | (o.mask << 24)), 0.0f, 0.0f};
    spec_own_store_products(row, o);
}
__device__ __forceinline__ SpecOwn spec_own_load(const spec_f4* row) {
    const spec_f4 q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3], q4 = row[4];
    SpecOwn o;
    o.p1 = __longlong_as_double((long long)(((unsigned long long)__float_as_uint(q4.y) << 32) | __float_as_uint(q4.x)));
    o.p2 = __longlong_as_double((long long)(((unsigned long long)__float_as_uint(q4.w) << 32) | __float_as_uint(q4.z)));
    o.theta = q0.x; o.s0 = q0.y; o.s1 = q0.z; o.s2 = q0.w;
    o.st = q1.x; o.a[0] = q1.y; o.b[0] = q1.z; o.a[1] = q1.w;
    o.b[1] = q2.x; o.pos[0] = __float_as_uint(q2.y); o.pos[1] = __float_as_uint(q2.z); o.k[0] = __float_as_uint(q2.w);
    o.k[1] = __float_as_uint(q3.x);
    const uint32_t meta = __float_as_uint(q3.y);
    o.tr[0] = meta & 0xFFu; o.tr[1] = (meta >> 8) & 0xFFu; o.n = (meta >> 16) & 0xFFu; o.mask = meta >> 24;
    return o;
}

// ---- the cross-rank sum INSIDE the in-kernel training loop (SPEC_WITH_EXCHANGE: a kernel variant of its own, compiled
//      when a multi-rank run first asks for it).  Monte-Carlo samples are sharded over the GPUs of a node; per
//      iteration every rank holds its loss sum, its non-finite count and, after the chain rule, one gradient sum per
//      parameter, and all ranks need the totals before the (replicated) optimizer step.  The protocol is the one-shot
//      direct-write exchange of collective.hip — same regions and abort word, its own area and call count, so launches of
//      that kernel and of this one can follow each other on the same bsvi_exchange — run by the OWNERS' WAVE alone: lane i owns parameter i
//      (SPEC_GENERIC_OWNERS == 0: every parameter has an owner in that one wave), so
//        1. lane i stores its parameter's sum (lane 0 also the two loss words) into this rank's row of EVERY region,
//        2. lane i reads every rank's entry of its parameter (and the loss words) in this rank's region until all carry the
//           call's number (bounded; anybody's abort ends it) and adds them in rank order: bit-identical totals on every rank.
//      No workgroup barrier: the other waves go on to the next iteration's first barrier (drawing its normals on the way),
//      where they wait for the owners to publish the new table as always.
#if defined(SPEC_WITH_EXCHANGE) && !SPEC_DIAG && !SPEC_ACCUMULATE_CHUNKS
#define SPEC_EXCHANGE 1
// An entry of the region's second area carries its value AND the number of the call that wrote it in one 8-byte word
// (xchg_ll_entry), stored with a single 64-bit store: a reader that finds the call's number has the value — no fence and no
// separate flag between "data written" and "data may be read", one store -> one load across the link instead of store, fence,
// flag, poll, load.  (Measured with one rank exchanging with itself, fine-grained memory: the flagged form of
// collective.hip's kernel added 2.5 us to a 4.8 us iteration.)
__device__ __forceinline__ unsigned long long* spec_xentry(unsigned char* region, uint32_t parity, uint32_t r, uint32_t capacity, uint32_t world) {
    return reinterpret_cast<unsigned long long*>(region + XCHG_HEADER_WORDS * 4 + (size_t)2 * world * capacity * 4) + ((size_t)parity * world + r) * capacity;
}
// Programs whose parameters do not all have an owner in one wave (more than 64 of them, or more than two table entries each:
// SPEC_GENERIC_OWNERS): the same entries, written and awaited PER THREAD — a tagged entry needs no rendezvous, so the thread
// that forms a parameter's sum stores it to every region and later polls that one entry of every rank, wherever in the
// workgroup it sits.  Every thread also polls the loss words: each owner gates its own step on the total's finiteness.
__device__ __forceinline__ void spec_xput(const SpecExchange* xg, uint32_t seq, uint32_t k, float v) {
    const uint32_t world = xg->world, rank = xg->rank, cap = xg->capacity, parity = seq & 1u;
    const unsigned long long e = ((unsigned long long)seq << 32) | __float_as_uint(v);
    for (uint32_t p = 0; p < world; ++p)
        __hip_atomic_store(spec_xentry(xg->peer[p], parity, rank, cap, world) + k, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool spec_xget(const SpecExchange* xg, uint32_t seq, uint32_t k, float& v) {
    const uint32_t world = xg->world, rank = xg->rank, cap = xg->capacity, parity = seq & 1u;
    unsigned char* const region = xg->peer[rank];
    uint32_t* const abort_word = reinterpret_cast<uint32_t*>(region) + XCHG_MAX_RANKS * XCHG_FLAG_STRIDE;
    unsigned long long t0 = 0;
    for (uint32_t round = 0;; ++round) {
        unsigned long long e[XCHG_MAX_RANKS];
#pragma unroll
        for (uint32_t r = 0; r < XCHG_MAX_RANKS; ++r) {
            e[r] = (unsigned long long)seq << 32;
            if (r < world) e[r] = __hip_atomic_load(spec_xentry(region, parity, r, cap, world) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        bool all = true;
        float t = 0.0f;
#pragma unroll
        for (uint32_t r = 0; r < XCHG_MAX_RANKS; ++r) {
            all = all && (uint32_t)(e[r] >> 32) == seq;
            if (r < world) t += __uint_as_float((uint32_t)e[r]);
        }
        if (all) {
            v = t;
            return __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u;
        }
        if (round == 0u) t0 = wall_clock64();
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;
        if (wall_clock64() - t0 > xg->timeout_ticks) {
            for (uint32_t p = 0; p < world; ++p)
                __hip_atomic_store(reinterpret_cast<uint32_t*>(xg->peer[p]) + XCHG_MAX_RANKS * XCHG_FLAG_STRIDE, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
#if !SPEC_GENERIC_OWNERS
// returns false when the call was abandoned (this rank or a peer gave up waiting: sticky, csrc/collective.hip)
__device__ __forceinline__ bool spec_exchange(const SpecExchange* xg, uint32_t seq, uint32_t l, bool has_param, float& vs, float& vb, float& gsum) {
    const uint32_t world = xg->world, rank = xg->rank, cap = xg->capacity, parity = seq & 1u;
    uint32_t* const abort_word = reinterpret_cast<uint32_t*>(xg->peer[rank]) + XCHG_MAX_RANKS * XCHG_FLAG_STRIDE;
    const uint32_t aborted = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);       // (needed at the end only)
    const unsigned long long tag = (unsigned long long)seq << 32;
    // 1. this rank's sums into its row of EVERY region (its own included): lane l its parameter's, lane 0 the loss words too
    for (uint32_t p = 0; p < world; ++p) {
        unsigned long long* const row = spec_xentry(xg->peer[p], parity, rank, cap, world);
        if (l == 0u) {
            __hip_atomic_store(row, tag | __float_as_uint(vs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(row + 1, tag | __float_as_uint(vb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (has_param) __hip_atomic_store(row + BSVI_OUT_HEADER + l, tag | __float_as_uint(gsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. every rank's row of THIS region, until all of them carry the call's number; added in rank order
    unsigned char* const region = xg->peer[rank];
    unsigned long long t0 = 0;
    bool gave_up = false;
    float ts, tb, tg;
    for (uint32_t round = 0;; ++round) {
        unsigned long long e0[XCHG_MAX_RANKS], e1[XCHG_MAX_RANKS], eg[XCHG_MAX_RANKS];
#pragma unroll
        for (uint32_t r = 0; r < XCHG_MAX_RANKS; ++r) {
            e0[r] = e1[r] = eg[r] = tag;
            if (r < world) {
                const unsigned long long* const row = spec_xentry(region, parity, r, cap, world);
                e0[r] = __hip_atomic_load(row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                e1[r] = __hip_atomic_load(row + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (has_param) eg[r] = __hip_atomic_load(row + BSVI_OUT_HEADER + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        bool all = true;
        ts = 0.0f; tb = 0.0f; tg = 0.0f;
#pragma unroll
        for (uint32_t r = 0; r < XCHG_MAX_RANKS; ++r) {
            all = all && (uint32_t)(e0[r] >> 32) == seq && (uint32_t)(e1[r] >> 32) == seq && (uint32_t)(eg[r] >> 32) == seq;
            if (r < world) { ts += __uint_as_float((uint32_t)e0[r]); tb += __uint_as_float((uint32_t)e1[r]); tg += __uint_as_float((uint32_t)eg[r]); }
        }
        if (__all((int)all)) break;
        // somebody is late: bounded, and ended by anybody's abort
        if (round == 0u) t0 = wall_clock64();
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { gave_up = true; break; }
        if (wall_clock64() - t0 > xg->timeout_ticks) {
            gave_up = true;
            if (l < world) __hip_atomic_store(reinterpret_cast<uint32_t*>(xg->peer[l]) + XCHG_MAX_RANKS * XCHG_FLAG_STRIDE, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (l == 0u) atomicAdd(abort_word + 1, 1u);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    if (gave_up || aborted != 0u) return false;
    vs = ts; vb = tb; gsum = tg;
    return true;
}
#endif
#else
#define SPEC_EXCHANGE 0
#endif

#if defined(SPEC_WAVES_PER_EU)
#define SPEC_VGPR_ATTR __attribute__((amdgpu_waves_per_eu(SPEC_WAVES_PER_EU, SPEC_WAVES_PER_EU)))
#elif defined(SPEC_NUM_VGPR)
#define SPEC_VGPR_ATTR __attribute__((amdgpu_num_vgpr(SPEC_NUM_VGPR)))
#else
#define SPEC_VGPR_ATTR
#endif
// SPEC_BOUND_THREADS (>= SPEC_MAX_THREADS): the launch bound the register allocator sees.  A bound of 768 / 1024 threads
// caps a lane at 168 / 128 registers, so that 3 / 4 workgroups of 256 threads share a CU (the throughput regime).
#ifndef SPEC_BOUND_THREADS
#define SPEC_BOUND_THREADS SPEC_MAX_THREADS
#endif
extern "C" __global__ void __launch_bounds__(SPEC_BOUND_THREADS) SPEC_VGPR_ATTR bsvi_spec_kernel(const SpecArgs A_unused) {
    (void)A_unused;
    const SPEC_CONST_AS char* ka = (const SPEC_CONST_AS char*)__builtin_amdgcn_kernarg_segment_ptr();
#define SPEC_A ((const SPEC_CONST_AS SpecArgs*)ka)
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x, W = nthreads >> 6, wave = tid >> 6, lane = tid & 63u;
    const uint32_t G = gridDim.x;
    float* const WS = spec_lds + SPEC_OFF_WS;
    float* const RED = spec_lds + SPEC_OFF_RED;
    float* const PS = spec_lds + SPEC_OFF_PS;
    uint32_t* const TAB = reinterpret_cast<uint32_t*>(spec_lds + SPEC_OFF_TAB);
    float* const WSw = WS + wave * SPEC_WS_PAD;

    SpecLane T;
    T.lane = lane;
    T.n = blockIdx.x * nthreads + tid;
    T.active = false; T.nc = 0; T.nidx = 0;
    T.vz = T.n >> 31;
    T.tile = wave * SPEC_TR_FLOATS + lane;
    const uint32_t sample_base = SPEC_A->sample_base;
    const uint32_t n_chunks = (SPEC_A->n_local + G * nthreads - 1u) / (G * nthreads);
    const uint32_t mode = SPEC_A->mode;
    const bool step = mode != SPEC_MODE_SUMS;
    const uint32_t n_it = (mode == SPEC_MODE_LOOP) ? SPEC_A->n_iterations : 1u;

    // ---- once per launch: tables, observed data, theta and the optimizer state into LDS; the uniform table
    {
        const uint32_t* uniform = reinterpret_cast<const uint32_t*>(SPEC_A->uniform);
        const uint32_t* pu_ptr = SPEC_A->pu_ptr;
        const uint32_t* pu_pos = SPEC_A->pu_pos;
        const uint32_t* pu_idx = SPEC_A->pu_idx;
        const float* obs = SPEC_A->obs;
        const float* params = SPEC_A->params;
        const float* state = SPEC_A->state;
        const uint8_t* mask = SPEC_A->mask;
        const uint8_t* mask_first = SPEC_A->mask_first;
        // The first trip of every copy loop is REQUESTED before any of them is stored: eight dependent global round trips
        // (~1 us each, the tables are cold) in a row were most of the launch's prologue — a third of a 20-iteration launch.
        const bool h_uni = tid < 4u * SPEC_N_UNIFORM, h_ptr = tid < SPEC_N_PARAMS + 1u, h_pos = tid < SPEC_N_POS,
                   h_obs = tid < SPEC_N_OBS, h_par = tid < SPEC_N_PARAMS;
        uint32_t r_uni = 0, r_ptr = 0, r_pos = 0, r_idx = 0, r_mask = 0;
        float r_obs = 0.0f, r_par = 0.0f, r_st[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const bool h_uni2 = tid + nthreads < 4u * SPEC_N_UNIFORM;            // (four words per entry: the table is the long one)
        uint32_t r_uni2 = 0;
        if (h_uni) r_uni = uniform[tid];
        if (h_uni2) r_uni2 = uniform[tid + nthreads];
        if (h_ptr) r_ptr = pu_ptr[tid];
        if (h_pos) { r_pos = pu_pos[tid]; r_idx = pu_idx[tid]; }
        if (h_obs) r_obs = obs[tid];
        if (h_par) {
            r_par = params[tid];
            if (step) {
                if (state) {
#pragma unroll
                    for (uint32_t s = 0; s < 4u; ++s) r_st[s] = state[(size_t)s * SPEC_N_PARAMS + tid];
                }
                r_mask = (mask[tid] ? 1u : 0u) | (mask_first[tid] ? 2u : 0u);
            }
        }
        asm volatile("" ::: "memory");        // (the loads stay in front of the stores: the compiler would pair them up again)
        if (h_uni) TAB[tid] = r_uni;
        if (h_uni2) TAB[tid + nthreads] = r_uni2;
        if (h_ptr) TAB[SPEC_TAB_PTR + tid] = r_ptr;
        if (h_pos) { TAB[SPEC_TAB_POS + tid] = r_pos; TAB[SPEC_TAB_IDX + tid] = r_idx; }
        if (h_obs) spec_lds[SPEC_N_UNIFORM + tid] = r_obs;
        if (h_par) {
            PS[tid] = r_par;
            if (step) {
#pragma unroll
                for (uint32_t s = 0; s < 4u; ++s) PS[(1u + s) * SPEC_NP_PAD + tid] = r_st[s];      // no state buffer: a fresh optimizer (all zeros), nothing written back
                TAB[SPEC_TAB_MASK + tid] = r_mask;
            }
        }
        for (uint32_t i = tid + 2u * nthreads; i < 4u * SPEC_N_UNIFORM; i += nthreads) TAB[i] = uniform[i];
        for (uint32_t i = tid + nthreads; i < SPEC_N_PARAMS + 1u; i += nthreads) TAB[SPEC_TAB_PTR + i] = pu_ptr[i];
        for (uint32_t j = tid + nthreads; j < SPEC_N_POS; j += nthreads) { TAB[SPEC_TAB_POS + j] = pu_pos[j]; TAB[SPEC_TAB_IDX + j] = pu_idx[j]; }
        for (uint32_t i = tid + nthreads; i < SPEC_N_OBS; i += nthreads) spec_lds[SPEC_N_UNIFORM + i] = obs[i];
        for (uint32_t i = tid + nthreads; i < SPEC_N_PARAMS; i += nthreads) {
            PS[i] = params[i];
            if (step) {
#pragma unroll
                for (uint32_t s = 0; s < 4u; ++s)
                    PS[(1u + s) * SPEC_NP_PAD + i] = state ? state[(size_t)s * SPEC_N_PARAMS + i] : 0.0f;
                TAB[SPEC_TAB_MASK + i] = (mask[i] ? 1u : 0u) | (mask_first[i] ? 2u : 0u);
            }
        }
        __syncthreads();
        const float* consts = SPEC_A->consts;
        for (uint32_t k = tid; k < SPEC_N_UNIFORM; k += nthreads) {
            const uint32_t src = TAB[4 * k], w1 = TAB[4 * k + 1];
            spec_publish_uniform(TAB, k, ((w1 >> 8) & 0xFFu) ? PS[src] : consts[src]);
        }
    }

    // the owner rows: built once, read back with four 16-byte LDS loads per iteration (kept in registers across the body
    // they are the first values the allocator spills to scratch: 2.2 us of scratch round trips per iteration)
    spec_f4* const OWN = reinterpret_cast<spec_f4*>(spec_lds + SPEC_OFF_OWN);
    // The owners are the first threads of wave SPEC_OWNER_WAVE when the workgroup has that wave and every parameter has an
    // owner (wave 1 by default: with five waves on four SIMDs wave 0 shares its SIMD with the fifth, and the serial epilogue
    // should not compete with that wave's work), else of wave 0.
#ifndef SPEC_OWNER_WAVE
#define SPEC_OWNER_WAVE 1u
#endif
    // The draw wave.  An iteration takes as long as the chain of the owners' wave: its draw, its body, the sums, the epilogue
    // (a single wave alone runs at nearly the speed of five, tools/wave_scaling.sh).  The in-kernel loop of a one-workgroup
    // launch is therefore given one wave more than the samples need (specialize.cpp, geo): it carries no samples and, beside
    // the epilogue, draws the NEXT iteration's normals of the owners' wave into its own — otherwise unused — transpose tile,
    // where the owners' wave picks them up behind the next barrier; every other wave draws beside the epilogue too, none
    // ahead of the barrier.  (Measured, tools/draw_wave_probe.sh: 64 samples 4.68 -> 3.83 us per iteration, 128: 4.79 -> 3.89,
    // 256: 4.91 -> 4.64; with five and more sample waves two of them share a SIMD and their late draws, not the owners'
    // chain, set the pace: 300 samples 5.0 -> 5.0.  The roles are a kernel variant of their own, SPEC_WITH_DRAW_WAVE, used
    // for up to four sample waves: merely compiled in they cost the plain loop 3 %.)  The draws are a function of (seed,
    // offset, sample index, row) only: who computes them does not change a bit of the result.
#if SPEC_TILE && SPEC_KEEP_NOISE && SPEC_KEEP_NOISE <= 68 && !SPEC_GENERIC_OWNERS && !SPEC_DIAG && !SPEC_ACCUMULATE_CHUNKS \
    && !defined(SPEC_DEBUG_NO_DRAW) && !defined(SPEC_NO_EARLY_DRAW) && !defined(SPEC_DEBUG_LATE_FREE) && defined(SPEC_WITH_DRAW_WAVE)
#define SPEC_DRAW_WAVE 1
    const uint32_t WSN = (SPEC_A->n_local + 63u) / 64u;     // waves that carry samples
    const bool has_draw_wave = mode == SPEC_MODE_LOOP && G == 1u && W >= 2u && WSN < W;
    const uint32_t fed_index = W == 2u ? 0u : 1u;         // the owners' wave, as without a draw wave (SPEC_OWNER_WAVE)
    // The draw SERVICE (round 5): five sample waves and three draw waves.  One wave issues a vector instruction every ~3.1 ns whether
    // or not a second wave shares its SIMD (profiles/r5/cfg1_wave_rate_notes.txt), so an iteration is the chain of one wave: body,
    // flush, sums — barrier — the owners' epilogue; with five sample waves the owners' wave also had to draw its own next noise in
    // front of that barrier (470 of its 1 650 instructions), and two late draws on one SIMD took longer than the epilogue they ran
    // beside (the Philox multiplies and the transcendentals are quarter-rate: a draw is 1.5 us of a SIMD's pipe).  Here NO sample wave
    // draws after the first iteration: the extra waves draw for all of them — draw wave k one set for sample wave k while the bodies
    // run (kept in registers, stored behind the barrier: a sample wave reads its buffer at the top of its body), and, when there are
    // more sample waves than draw waves, a second set beside the epilogue unless it sits on the owners' SIMD — into one buffer per
    // sample wave in the draw waves' own, otherwise unused, transpose tiles.  (300 samples: waves 5, 6, 7 for 0 | 1, 3 | 2, 4.)
    // The draws are functions of (seed, offset, sample, row): who computes them does not change a bit.
    const bool draw_service = has_draw_wave && W - WSN >= 3u;      // (three or more extra waves: the host asked for the service)
    const bool draw_wave = __builtin_amdgcn_readfirstlane((has_draw_wave && (draw_service ? wave >= WSN : wave == W - 1u)) ? 1 : 0) != 0;
    const bool fed_wave = __builtin_amdgcn_readfirstlane((has_draw_wave && (draw_service ? wave < WSN : wave == fed_index)) ? 1 : 0) != 0;
    // (one wave's set: SPEC_KEEP_NOISE rows of 64 lanes; the service's five buffers lie in the three draw waves' tiles)
    float* const NZ = spec_lds + SPEC_OFF_TR + (draw_service ? WSN * SPEC_TR_FLOATS + wave * (SPEC_KEEP_NOISE * 64u) : (W - 1u) * SPEC_TR_FLOATS) + lane;
    float* const NZB = spec_lds + SPEC_OFF_TR + WSN * SPEC_TR_FLOATS + lane;           // (service: buffer w at NZB + w * SPEC_KEEP_NOISE * 64)
    if (draw_wave) for (uint32_t k = lane; k < SPEC_WS_PAD; k += 64u) WSw[k] = 0.0f;     // its row of sums stays zero
#else
#define SPEC_DRAW_WAVE 0
    const bool has_draw_wave = false, draw_wave = false, fed_wave = false;
    const uint32_t fed_index = 1u;
#endif
    const uint32_t own_base = has_draw_wave ? fed_index * 64u
                            : (!SPEC_GENERIC_OWNERS && SPEC_OWNER_WAVE * 64u + SPEC_N_PARAMS <= nthreads) ? SPEC_OWNER_WAVE * 64u : 0u;
    const uint32_t oid = tid - own_base;                      // the parameter this thread owns (if < SPEC_N_PARAMS)
    bool own_fast = false;
    if (oid < SPEC_N_PARAMS) {
        const uint32_t j0 = TAB[SPEC_TAB_PTR + oid], j1 = TAB[SPEC_TAB_PTR + oid + 1];
        if (j1 - j0 <= 2u) {
            own_fast = true;
            SpecOwn own;
            own.n = j1 - j0;
            own.mask = 0;
            own.theta = PS[oid];
            own.s0 = own.s1 = own.s2 = own.st = 0.0f;
            own.p1 = own.p2 = 1.0;
            own.a[0] = own.a[1] = own.b[0] = own.b[1] = 0.0f;
            own.pos[0] = own.pos[1] = own.k[0] = own.k[1] = own.tr[0] = own.tr[1] = 0;
            if (step) {
                own.s0 = PS[SPEC_NP_PAD + oid]; own.s1 = PS[2 * SPEC_NP_PAD + oid];
                own.s2 = PS[3 * SPEC_NP_PAD + oid]; own.st = PS[4 * SPEC_NP_PAD + oid];
                own.mask = TAB[SPEC_TAB_MASK + oid];
                const bsvi_opt_cfg cfg0 = SPEC_A->cfg;       // once per launch: the running products start at beta^st
                if (cfg0.kind != BSVI_OPT_SGD && own.st != 0.0f) {
                    own.p1 = pow((double)cfg0.beta1, (double)own.st);
                    own.p2 = pow((double)cfg0.beta2, (double)own.st);
                }
            }
#pragma unroll
            for (uint32_t e = 0; e < 2u; ++e) {
                if (e < own.n) {
                    const uint32_t k = TAB[SPEC_TAB_IDX + j0 + e];
                    own.pos[e] = TAB[SPEC_TAB_POS + j0 + e];
                    own.k[e] = k;
                    own.tr[e] = TAB[4 * k + 1] & 0xFFu;
                    own.a[e] = __uint_as_float(TAB[4 * k + 2]);
                    own.b[e] = __uint_as_float(TAB[4 * k + 3]);
                }
            }
            spec_own_store(OWN + 5 * oid, own);
        }
    }

    // ---- what the iterations read of the argument block, once: a scalar load per iteration and phase is a round trip
    //      to the scalar cache on the critical path of a ~5 us iteration
    SPEC_RELOAD_ARGS();
    SpecBody B0;
    B0.n_local = SPEC_A->n_local;
    B0.seed_lo = SPEC_A->seed_lo;
    B0.seed_hi = SPEC_A->seed_hi;
#if SPEC_DIAG
    B0.noise = SPEC_A->noise;
    B0.samples_out = SPEC_A->samples_out;
    B0.noise_out = SPEC_A->noise_out;
    B0.fvalue_out = SPEC_A->fvalue_out;
    B0.f_weight = SPEC_A->f_weight;
    B0.q_weight = SPEC_A->q_weight;
#else
    B0.noise = nullptr; B0.samples_out = nullptr; B0.noise_out = nullptr; B0.fvalue_out = nullptr;
    B0.f_weight = nullptr; B0.q_weight = nullptr;
#endif
    unsigned long long off0 = ((unsigned long long)SPEC_A->offset_hi << 32) | SPEC_A->offset_lo;
    if (const unsigned long long* const offset_dev = SPEC_A->offset_dev) off0 += *offset_dev;
    const uint32_t n_global = SPEC_A->n_global;
    const uint32_t pretraining = SPEC_A->pretraining_iterations;
    // several workgroups in loop mode (the many-workgroup geometry only: the one-workgroup kernels are never launched with
    // more, and the extra live scalars cost BASELINE config 1's kernel 30 more spilled scalar registers).  The arrival ticket
    // and the generation number of the launch are ZERO when it starts (the host clears both words on the launch's stream,
    // specialize.cpp): iteration `it` is released as generation it + 1, and SPEC_GEN_OVER says "this launch is over" — a
    // workgroup that only becomes resident after workgroup 0 gave up on it leaves without touching the ticket.
#if SPEC_ACCUMULATE_CHUNKS
#define SPEC_LOOP_MANY 1
#define SPEC_GEN_OVER 0x7fff0000u
    const unsigned int gen0 = 0;
    if (G > 1 && mode == SPEC_MODE_LOOP && blockIdx.x != 0 &&
        __hip_atomic_load(SPEC_A->ticket + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= SPEC_GEN_OVER) return;
#else
#define SPEC_LOOP_MANY 0
#endif
#if SPEC_EXCHANGE
    // (this rank's call count lives in its region and is touched by this rank's kernels only: stream order makes it current)
    uint32_t xseq0 = 0;
    if (const SpecExchange* const xg0 = SPEC_A->xchg) xseq0 = reinterpret_cast<const uint32_t*>(xg0->peer[xg0->rank])[XCHG_CALLS_WORD + 1];
#endif

#if defined(SPEC_DEBUG_STAMPS)        // timing experiment (tools/spec_stamps.py): s_memtime at the phase boundaries of one iteration
    unsigned long long stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SPEC_STAMP(i) if (it == n_it / 2) stamp[i] = __builtin_amdgcn_s_memtime()
#else
#define SPEC_STAMP(i)
#endif
    // One workgroup, noise kept in registers, lean build: the NEXT iteration's standard normals do not depend on anything
    // this iteration computes, and most waves have idle time right after their body — N = 300 is five waves on four SIMDs,
    // the hardware favours the older wave of a SIMD, so waves 0-3 wait at the barrier below for wave 4 — so the first wave
    // of each SIMD draws its next noise THERE.  What is left after the barrier is the owners' epilogue (threads of wave 0)
    // with only the late waves' draws beside it, instead of the epilogue followed by wave 0's own draw.
#if SPEC_KEEP_NOISE && !SPEC_DIAG && !SPEC_ACCUMULATE_CHUNKS && !defined(SPEC_DEBUG_NO_DRAW) && !defined(SPEC_NO_EARLY_DRAW)
#define SPEC_EARLY_DRAW 1
    SpecNoise Z;                       // lives across iterations: the early draw overwrites it once the body is done with it
    bool noise_ready = false;
#else
#define SPEC_EARLY_DRAW 0
#endif
#if SPEC_DRAW_WAVE
    if (draw_service && draw_wave) {
        // The service's draw waves run a loop of their OWN with the main loop's two barriers per iteration (a set drawn beside the
        // bodies would otherwise be live across spec_body in the register allocation of EVERY wave: 15 spilled registers and
        // 4.7 -> 6.5 us per iteration for every user of this kernel variant).  Their rows of the sums stay zero.
        if (lane == 0u) { RED[8 + 2 * wave] = 0.0f; RED[9 + 2 * wave] = 0.0f; }
        const uint32_t n_draw = W - WSN, k = wave - WSN;
        const uint32_t first = k;                              // (k < WSN or nothing: `has_first`)
        uint32_t before = 0;                                   // draw waves in front of this one that take a second set
        for (uint32_t j = 0; j < k; ++j) before += (((WSN + j) & 3u) != 1u) ? 1u : 0u;
        const uint32_t second = n_draw + before;
        const bool has_first = first < WSN, has_second = (wave & 3u) != 1u && second < WSN;      // (wave & 3 == 1: the owners' SIMD)
        auto draw_for = [&](uint32_t target, unsigned long long offn, SpecNoise& Zd) {
            SpecLane Tn = T;
            Tn.n = target * 64u + lane;
            Tn.active = Tn.n < B0.n_local;
            Tn.nc = Tn.active ? Tn.n : (B0.n_local - 1u);
            Tn.nidx = sample_base + Tn.nc;
            Tn.off_lo = (uint32_t)offn;
            Tn.off_hi = (uint32_t)(offn >> 32);
            Tn.vz = T.vz;
            spec_draw(B0, Tn, Zd);
        };
        for (uint32_t it = 0; it < n_it; ++it) {
            const bool more = it + 1u < n_it;
            spec_lds_barrier();                                // (the main loop's first barrier: the sample waves read their buffers behind it)
            SpecNoise Za;
            if (more && has_first) draw_for(first, off0 + it + 1u, Za);     // beside the bodies, in registers
            spec_lds_barrier();                                // (the second: the buffers are free)
            if (more) {
                if (has_first) {
#pragma unroll
                    for (uint32_t r = 0; r < SPEC_KEEP_NOISE; ++r) NZB[first * (SPEC_KEEP_NOISE * 64u) + 64u * r] = Za.z[r];
                }
                if (has_second) {
                    SpecNoise Zb;
                    draw_for(second, off0 + it + 1u, Zb);
#pragma unroll
                    for (uint32_t r = 0; r < SPEC_KEEP_NOISE; ++r) NZB[second * (SPEC_KEEP_NOISE * 64u) + 64u * r] = Zb.z[r];
                }
            }
        }
        return;
    }
#endif
    for (uint32_t it = 0; it < n_it; ++it) {
        SPEC_STAMP(0);
        // ---- one Monte-Carlo sample per lane and chunk (a workgroup of a large shard walks several chunks of samples, so
        //      that the launch's fixed costs — tables, prologue, the row of sums — are paid once per workgroup, not once
        //      per 256 samples).  The standard normals do not depend on the uniform table: the first chunk's are drawn
        //      ahead of the barrier that publishes it.
        float lane_value = 0.0f, lane_bad = 0.0f;
        for (uint32_t chunk = 0; chunk < n_chunks; ++chunk) {
            SpecBody B = B0;
#if SPEC_DIAG
            // a given-noise sequence is laid out [iteration][row][n_local]
            if (B.noise) B.noise += (size_t)it * SPEC_N_NOISE * B.n_local;
#endif
            T.n = (chunk * G + blockIdx.x) * nthreads + tid;
            T.active = T.n < B.n_local;
            T.nc = T.active ? T.n : (B.n_local - 1u);
            T.nidx = sample_base + T.nc;
            const unsigned long long off = off0 + it;
            T.f = 0.0f;
            T.lq = 0.0f;
#if SPEC_DIAG
            T.gw = B.f_weight ? B.f_weight[T.nc] : 1.0f;
#else
            T.gw = 1.0f;
#endif
            T.off_lo = (uint32_t)off;
            T.off_hi = (uint32_t)(off >> 32);
#if !SPEC_EARLY_DRAW
            SpecNoise Z;
#endif
#if defined(SPEC_DEBUG_NO_DRAW)                                // timing experiment (BSVI_SPEC_DEFINES): what the noise costs
            for (uint32_t r = 0; r < (SPEC_KEEP_NOISE ? SPEC_KEEP_NOISE : 1); ++r) Z.z[r] = 0.25f;
#elif SPEC_EARLY_DRAW
#if defined(SPEC_DEBUG_LATE_FREE)                              // timing experiment: what the late waves' own draw costs
            if (!noise_ready && (!((SPEC_DEBUG_LATE_FREE >> wave) & 1u) || it == 0u)) spec_draw(B, T, Z);   // mask of waves whose draws are skipped
#else
            if (!noise_ready && !draw_wave && !(fed_wave && it > 0u)) spec_draw(B, T, Z);
#endif
#else
            spec_draw(B, T, Z);
#endif
            SPEC_STAMP(1);
            if (chunk == 0) {
                spec_lds_barrier();                            // the uniform table of this iteration is complete ...
#if SPEC_ACCUMULATE_CHUNKS
                for (uint32_t k = lane; k < SPEC_WS_CELLS; k += 64u) WSw[k] = 0.0f;     // ... and the last one's sums are consumed
#endif
            }
#if SPEC_DRAW_WAVE
            if (fed_wave && it > 0u) {                         // drawn by the draw wave beside the last epilogue
#pragma unroll
                for (uint32_t r = 0; r < SPEC_KEEP_NOISE; ++r) Z.z[r] = NZ[64u * r];
            }
#endif
            SPEC_STAMP(2);
#if !defined(SPEC_DEBUG_NO_BODY)
            if (!draw_wave) {
                spec_body(B, T, Z, WSw);
#if SPEC_TILE
                spec_du_flush(spec_lds + SPEC_OFF_TR + wave * SPEC_TR_FLOATS, WSw, lane);
#endif
            }
#endif
            SPEC_STAMP(3);
            const float value = (SPEC_ESTIMATOR == BSVI_EST_BLACKBOX) ? (T.lq * T.f + T.f) : T.f;
            lane_value += T.active ? value : 0.0f;
            lane_bad += (T.active && !isfinite(value)) ? 1.0f : 0.0f;
        }
        const float vsum = wave_sum(lane_value);
        const float bad = wave_sum(lane_bad);
        if (lane == 0) { RED[8 + 2 * wave] = vsum; RED[9 + 2 * wave] = bad; }
        SPEC_STAMP(4);
#if SPEC_EARLY_DRAW
        noise_ready = false;
#ifndef SPEC_EARLY_MASK
#define SPEC_EARLY_MASK (1u | (1u << SPEC_OWNER_WAVE))          // wave 0 (idle at the barrier) and the owners' wave (busy after it)
#endif
#if defined(SPEC_DEBUG_LATE_FREE)
        if (((SPEC_EARLY_MASK >> wave) & 1u) && !((SPEC_DEBUG_LATE_FREE >> wave) & 1u) && n_chunks == 1u && it + 1u < n_it) {
#else
        if (((SPEC_EARLY_MASK >> wave) & 1u) && !has_draw_wave && n_chunks == 1u && it + 1u < n_it) {
#endif
            const unsigned long long off = off0 + it + 1u;
            SpecLane Tn = T;
            Tn.off_lo = (uint32_t)off;
            Tn.off_hi = (uint32_t)(off >> 32);
            spec_draw(B0, Tn, Z);
            noise_ready = true;
        }
#endif
        spec_lds_barrier();                                    // every wave's sums are in WS / RED
        SPEC_STAMP(5);
#if SPEC_DRAW_WAVE
        if (draw_wave && it + 1u < n_it) {                     // (the owners' wave read the last set behind the barrier before the body)
            const unsigned long long off = off0 + it + 1u;
            SpecLane Tn = T;
            Tn.n = fed_index * 64u + lane;
            Tn.active = Tn.n < B0.n_local;
            Tn.nc = Tn.active ? Tn.n : (B0.n_local - 1u);
            Tn.nidx = sample_base + Tn.nc;
            Tn.off_lo = (uint32_t)off;
            Tn.off_hi = (uint32_t)(off >> 32);
            SpecNoise Zf;
            spec_draw(B0, Tn, Zf);
#pragma unroll
            for (uint32_t r = 0; r < SPEC_KEEP_NOISE; ++r) NZ[64u * r] = Zf.z[r];
        }
#endif

        // ---- several workgroups: every one publishes its row of sums, the last to arrive adds the rows in order
        uint32_t rows = W;                                     // rows of WS / RED that hold sums
        if (G > 1) {
            SPEC_RELOAD_ARGS();
            float* const partials = SPEC_A->partials;
            unsigned int* const ticket = SPEC_A->ticket;
            const uint32_t stride = 2u + SPEC_N_POS;
            float* mine = partials + (size_t)blockIdx.x * stride;
            for (uint32_t k = tid; k < SPEC_N_POS; k += nthreads) mine[2 + k] = spec_pos_total(WS, k, W);
            if (tid == 0) {
                float s = 0.0f, c = 0.0f;
                for (uint32_t w = 0; w < W; ++w) { s += RED[8 + 2 * w]; c += RED[9 + 2 * w]; }
                mine[0] = s;
                mine[1] = c;
            }
            __syncthreads();                                   // every wave's stores issued and drained
#if SPEC_LOOP_MANY
            if (mode == SPEC_MODE_LOOP) {
                // The loop of inference.py:95-108 over SEVERAL workgroups in one launch (round 4; before: one launch per
                // iteration, every workgroup staging the tables again — 70 x the algorithmic traffic at BASELINE config 2).
                // Workgroup 0 is the iteration's owner: it waits for the other G - 1 rows, adds them, runs the epilogue on ITS
                // copy of the parameters and the optimizer state, writes the new parameters to memory and releases the
                // iteration's generation number; the others wait for that number, read the parameters back and rebuild their
                // copy of the uniform table.  Generation numbers only grow (base read at the start of the launch), the waits
                // are bounded (a launch whose workgroups are not all resident must not hang: it ends with a NaN loss).
                unsigned int* const gen = ticket + 64;
                const unsigned int target = gen0 + it + 1u;
                if (blockIdx.x != 0) {
                    if (tid == 0) {
                        __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long t0 = wall_clock64();
                        unsigned int seen;
                        bool over = false;
                        while ((int)((seen = __hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) - target) < 0) {
                            if (wall_clock64() - t0 > 400000000ull) { over = true; break; }      // 4 s of the 100 MHz clock
                            __builtin_amdgcn_s_sleep(2);
                        }
                        // (workgroup 0 gave up on a workgroup and ended the launch, or this one waited in vain: leave — going on
                        //  would mean K more waits of four seconds)
                        RED[3] = (over || seen >= SPEC_GEN_OVER) ? 1.0f : 0.0f;
                    }
                    __syncthreads();
                    if (RED[3] != 0.0f) return;
                    if (it + 1u == n_it) return;
                    float* const params = SPEC_A->params;
                    for (uint32_t i = tid; i < SPEC_N_PARAMS; i += nthreads)
                        PS[i] = __hip_atomic_load(params + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    const float* const consts = SPEC_A->consts;
                    (void)consts;
                    for (uint32_t k = tid; k < SPEC_N_UNIFORM; k += nthreads) {
                        const uint32_t src = TAB[4 * k], w1 = TAB[4 * k + 1];
                        if ((w1 >> 8) & 0xFFu) spec_publish_uniform(TAB, k, PS[src]);
                    }
                    continue;                                  // (the next iteration's first barrier publishes the table)
                }
                if (tid == 0) {
                    const unsigned long long t0 = wall_clock64();
                    bool late = false;
                    while (__hip_atomic_load(ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != G - 1u) {
                        if (wall_clock64() - t0 > 400000000ull) { late = true; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    RED[3] = late ? 1.0f : 0.0f;
                }
                __syncthreads();
                if (RED[3] != 0.0f) {
                    // A workgroup never arrived (not resident: a shared GPU, a CU mask, another stream's kernels).  The launch ends
                    // HERE: this and every remaining iteration get a NaN loss and no step (the parameters in memory are those of
                    // the last good iteration), the output block says so, and the generation number is set to "over" so that
                    // every waiting workgroup — and any that starts later — leaves at once.  (Round 4 went on to the next
                    // iteration and waited four seconds again: a K-iteration call took K x 4 s to fail.)
                    SPEC_RELOAD_ARGS();
                    float* const loss_slot = SPEC_A->loss_slot;
                    float* const finite_slot = SPEC_A->finite_slot;
                    float* const out = SPEC_A->out;
                    const float nanv = __int_as_float(0x7fc00000);
                    for (uint32_t k = it + tid; k < n_it; k += nthreads) {
                        if (loss_slot) loss_slot[k] = nanv;
                        if (finite_slot) finite_slot[k] = 0.0f;
                    }
                    if (tid == 0) {
                        out[0] = nanv; out[1] = 1.0f; out[2] = nanv; out[3] = 0.0f;
                        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(gen, SPEC_GEN_OVER, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return;
                }
            } else
#endif
            {
                if (tid == 0) {
                    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    RED[2] = (t == G - 1u) ? 1.0f : 0.0f;
                }
                __syncthreads();
                if (RED[2] == 0.0f) return;
            }
            // column c of the G rows: four interleaved slices per column (G can be hundreds of workgroups), then the slices
            float* const SL = spec_lds + SPEC_OFF_SCR;
            for (uint32_t i = tid; i < 4u * stride; i += nthreads) {
                const uint32_t sl = i & 3u, c = i >> 2;
                // eight loads in flight, added in row order (the same association as one load at a time: with hundreds of
                // workgroups the one-at-a-time form made this the longest phase of the launch — a serial tail in ONE
                // workgroup of ~130 dependent L2 round trips while the rest of the chip was idle)
                float s = 0.0f;
                uint32_t b = sl;
                for (; b + 28u < G; b += 32u) {
                    float v[8];
#pragma unroll
                    for (uint32_t u = 0; u < 8u; ++u)
                        v[u] = __hip_atomic_load(&partials[(size_t)(b + 4u * u) * stride + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (uint32_t u = 0; u < 8u; ++u) s += v[u];
                }
                for (; b < G; b += 4u)
                    s += __hip_atomic_load(&partials[(size_t)b * stride + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                SL[i] = s;
            }
            __syncthreads();
            for (uint32_t c = tid; c < stride; c += nthreads) {
                const float s = (SL[4 * c] + SL[4 * c + 1]) + (SL[4 * c + 2] + SL[4 * c + 3]);
                if (c < 2u) {
                    RED[8 + c] = s;
                } else {                                       // the grid total in the first wave's cell of the position
                    WS[SPEC_WS_CELL(c - 2u)] = s;
#if !SPEC_TILE
                    WS[4u * (c - 2u) + 1u] = 0.0f; WS[4u * (c - 2u) + 2u] = 0.0f; WS[4u * (c - 2u) + 3u] = 0.0f;
#endif
                }
            }
            if (tid == 0) {
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            rows = 1;
        }

        // ---- epilogue.  The thread that owns parameter i does everything that depends on it: adds the waves' sums of
        //      its uniform entries, applies the chain rule, the optimizer step on the LDS working copy, and publishes
        //      the entries' new values for the next iteration — so an iteration of the in-kernel loop has two
        //      barriers: after the body, and before the next one.
        // (the epilogue's own words of the argument block: requested here, they arrive while the LDS reads below are in flight)
        SPEC_RELOAD_ARGS();
        float* const out = SPEC_A->out;
        float* const loss_slot = SPEC_A->loss_slot;
        float* const finite_slot = SPEC_A->finite_slot;
        const bsvi_opt_cfg cfg = SPEC_A->cfg;
        // (literal trip counts and literal addresses — every row of RED exists, the rows of waves this launch does not have are
        //  read and not selected: the LDS reads are one base register + immediates and issue back to back)
        float vs = 0.0f, vb = 0.0f;
#pragma unroll
        for (uint32_t w = 0; w < SPEC_MAX_WAVES; ++w) {
            const float a = RED[8 + 2 * w], b = RED[9 + 2 * w];
            vs += w < rows ? a : 0.0f;
            vb += w < rows ? b : 0.0f;
        }
#if SPEC_EXCHANGE
        // several ranks: the owners' wave exchanges [loss sum, non-finite count, gradient sum per parameter] with its peers
        // (spec_exchange above); everything below runs on the TOTALS, as bsvi_finalize_step does behind the exchange kernel
        const SpecExchange* const xg = SPEC_A->xchg;
#if SPEC_GENERIC_OWNERS
        const bool xrun = xg != nullptr && mode == SPEC_MODE_LOOP;
        float xgsum = 0.0f;
        if (xrun) {                                            // per thread: see spec_xput / spec_xget
            const uint32_t xseq = xseq0 + it + 1u;
            if (tid == own_base) { spec_xput(xg, xseq, 0u, vs); spec_xput(xg, xseq, 1u, vb); }
            if (own_fast) {
                const SpecOwn own = spec_own_load(OWN + 5 * oid);
                float g = 0.0f;
#pragma unroll
                for (uint32_t e = 0; e < 2u; ++e) {
                    const float term = spec_pos_total(WS, own.pos[e], rows) * (own.b[e] * spec_utransform_grad(own.tr[e], own.theta));
                    g += e < own.n ? term : 0.0f;
                }
                spec_xput(xg, xseq, BSVI_OUT_HEADER + oid, g);
            }
            for (uint32_t i = tid; i < SPEC_N_PARAMS; i += nthreads) {
                if (i == oid && own_fast) continue;
                const float theta = PS[i];
                const uint32_t j0 = TAB[SPEC_TAB_PTR + i], j1 = TAB[SPEC_TAB_PTR + i + 1];
                float g = 0.0f;
                for (uint32_t j = j0; j < j1; ++j) {
                    const uint32_t pos = TAB[SPEC_TAB_POS + j], k = TAB[SPEC_TAB_IDX + j];
                    g += spec_pos_total(WS, pos, rows) * (__uint_as_float(TAB[4 * k + 3]) * spec_utransform_grad(TAB[4 * k + 1] & 0xFFu, theta));
                }
                spec_xput(xg, xseq, BSVI_OUT_HEADER + i, g);
            }
            bool ok = spec_xget(xg, xseq, 0u, vs);
            ok = spec_xget(xg, xseq, 1u, vb) && ok;
            if (own_fast) ok = spec_xget(xg, xseq, BSVI_OUT_HEADER + oid, xgsum) && ok;
            if (!ok) vs = __int_as_float(0x7fc00000);         // abandoned: NaN loss, no step
            if (tid == own_base && it + 1u == n_it) reinterpret_cast<uint32_t*>(xg->peer[xg->rank])[XCHG_CALLS_WORD + 1] = xseq0 + n_it;
        }
#else
        const bool xrun = xg != nullptr && mode == SPEC_MODE_LOOP && (tid >> 6) == (own_base >> 6);
        float xgsum = 0.0f;
        if (xrun) {
            if (own_fast) {
                const SpecOwn own = spec_own_load(OWN + 5 * oid);
                float tot[2];
#pragma unroll
                for (uint32_t e = 0; e < 2u; ++e) tot[e] = spec_pos_total(WS, own.pos[e], rows);
#pragma unroll
                for (uint32_t e = 0; e < 2u; ++e) {
                    const float term = tot[e] * (own.b[e] * spec_utransform_grad(own.tr[e], own.theta));
                    xgsum += e < own.n ? term : 0.0f;
                }
            }
            if (!spec_exchange(xg, xseq0 + it + 1u, oid, own_fast, vs, vb, xgsum)) vs = __int_as_float(0x7fc00000);     // abandoned: NaN loss, no step
            if (oid == 0u && it + 1u == n_it) reinterpret_cast<uint32_t*>(xg->peer[xg->rank])[XCHG_CALLS_WORD + 1] = xseq0 + n_it;
        }
#endif
#endif
        // -vs / n is finite exactly when vs is (n >= 1): the optimizer step does not wait for the division
        const float finite = isfinite(vs) ? 1.0f : 0.0f;
        const float loss = -vs / (float)n_global;
        SPEC_STAMP(7);
        if (tid == own_base) {
            if (it + 1u == n_it) { out[0] = vs; out[1] = vb; }     // (the output block of the launch's last iteration)
            if (step) {
                if (it + 1u == n_it) { out[2] = loss; out[3] = finite; }
                if (loss_slot) loss_slot[it] = loss;
                if (finite_slot) finite_slot[it] = finite;
            }
        }
        const float scale = step ? -1.0f / (float)n_global : 1.0f;
        const bool last = it + 1u == n_it;
        const uint32_t mask_bit = (mode == SPEC_MODE_LOOP && it <= pretraining) ? 2u : 1u;
        if (own_fast) {
            SpecOwn own = spec_own_load(OWN + 5 * oid);
            // (both positions' rows requested before either sum: an unused entry has position 0, read and not selected)
            float tot[2];
#pragma unroll
            for (uint32_t e = 0; e < 2u; ++e) tot[e] = spec_pos_total(WS, own.pos[e], rows);
            float gsum = 0.0f;
#pragma unroll
            for (uint32_t e = 0; e < 2u; ++e) {
                const float term = tot[e] * (own.b[e] * spec_utransform_grad(own.tr[e], own.theta));
                gsum += e < own.n ? term : 0.0f;
            }
#if SPEC_EXCHANGE
            if (xrun) gsum = xgsum;
#endif
            const float grad = gsum * scale;
            SPEC_STAMP(8);
            if (last || !step) out[BSVI_OUT_HEADER + oid] = grad;
            if (step) {
                if (finite != 0.0f && (own.mask & mask_bit))
                    optimizer_apply_running(cfg, own.theta, own.s0, own.s1, own.s2, own.st, grad, own.p1, own.p2);
                SPEC_STAMP(9);
#if SPEC_LOOP_MANY
                if (G > 1 && !last) __hip_atomic_store(SPEC_A->params + oid, own.theta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                if (last) {
                    float* const params = SPEC_A->params;
                    float* const state = SPEC_A->state;
                    params[oid] = own.theta;
                    if (state) {
                        state[oid] = own.s0;
                        state[(size_t)SPEC_N_PARAMS + oid] = own.s1;
                        state[2 * (size_t)SPEC_N_PARAMS + oid] = own.s2;
                        state[3 * (size_t)SPEC_N_PARAMS + oid] = own.st;
                    }
                } else {
#pragma unroll
                    for (uint32_t e = 0; e < 2u; ++e)
                        if (e < own.n) spec_store_uniform(own.k[e], own.a[e] + own.b[e] * utransform_common(own.tr[e], own.theta));
                    OWN[5 * oid] = spec_f4{own.theta, own.s0, own.s1, own.s2};
                    spec_lds[SPEC_OFF_OWN + SPEC_OWN_WORDS * oid + 4] = own.st;
                    if (cfg.kind != BSVI_OPT_SGD) spec_own_store_products(OWN + 5 * oid, own);
                }
            }
        }
#if SPEC_GENERIC_OWNERS      // parameters beyond the workgroup size or with more than two uniform entries: the LDS working copy
        for (uint32_t i = tid; i < SPEC_N_PARAMS; i += nthreads) {
            if (i == oid && own_fast) continue;
            const float theta = PS[i];
            const uint32_t j0 = TAB[SPEC_TAB_PTR + i], j1 = TAB[SPEC_TAB_PTR + i + 1];
            float gsum = 0.0f;
            for (uint32_t j = j0; j < j1; ++j) {
                const uint32_t pos = TAB[SPEC_TAB_POS + j], k = TAB[SPEC_TAB_IDX + j];
                gsum += spec_pos_total(WS, pos, rows) * (__uint_as_float(TAB[4 * k + 3]) * spec_utransform_grad(TAB[4 * k + 1] & 0xFFu, theta));
            }
#if SPEC_EXCHANGE
            bool xok = true;
            if (xrun) xok = spec_xget(xg, xseq0 + it + 1u, BSVI_OUT_HEADER + i, gsum);      // the ranks' sums of this parameter, in rank order
#else
            const bool xok = true;
#endif
            const float grad = gsum * scale;
            if (last || !step) out[BSVI_OUT_HEADER + i] = grad;
            if (step && xok && finite != 0.0f && (TAB[SPEC_TAB_MASK + i] & mask_bit))
                optimizer_update(cfg, PS, PS + SPEC_NP_PAD, SPEC_NP_PAD, i, grad);
            if (!step) continue;
            if (last) {
                float* const params = SPEC_A->params;
                float* const state = SPEC_A->state;
                params[i] = PS[i];
                if (state) {
#pragma unroll
                    for (uint32_t s = 0; s < 4u; ++s) state[(size_t)s * SPEC_N_PARAMS + i] = PS[(1u + s) * SPEC_NP_PAD + i];
                }
            } else {
                const float theta2 = PS[i];
                for (uint32_t j = j0; j < j1; ++j) spec_publish_uniform(TAB, TAB[SPEC_TAB_IDX + j], theta2);
#if SPEC_LOOP_MANY
                if (G > 1) __hip_atomic_store(SPEC_A->params + i, theta2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            }
        }
#endif
#if SPEC_LOOP_MANY
        if (G > 1 && mode == SPEC_MODE_LOOP) {                  // (only workgroup 0 gets here) the new parameters are out: release the iteration
            __threadfence();
            __syncthreads();
            if (tid == 0) __hip_atomic_store(SPEC_A->ticket + 64, gen0 + it + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        SPEC_STAMP(6);
    }
#if defined(SPEC_DEBUG_STAMPS)
    float* const loss_slot = SPEC_A->loss_slot;
    if (tid == 0 && loss_slot && n_it > 16) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int i = 1; i < 12; ++i) loss_slot[i] = (float)(stamp[i] - stamp[0]);
    }
#endif
#undef SPEC_A
}

}  // namespace bsvi
