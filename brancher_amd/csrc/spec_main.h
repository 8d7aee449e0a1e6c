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
#define SPEC_TAB_IDX (SPEC_TAB_POS + SPEC_N_UGRAD)
#define SPEC_TAB_MASK (SPEC_TAB_IDX + SPEC_N_UGRAD)

// U[k] = a + b * g(x) and its companions, from the LDS copy of entry k
__device__ __forceinline__ void spec_publish_uniform(const uint32_t* TAB, uint32_t k, float x) {
    const uint32_t w1 = TAB[4 * k + 1];
    const float u = __uint_as_float(TAB[4 * k + 2]) + __uint_as_float(TAB[4 * k + 3]) * utransform((int)(w1 & 0xFFu), x);
    spec_lds[k] = u;
    spec_lds[SPEC_OFF_UR + k] = 1.0f / u;
    spec_lds[SPEC_OFF_UL + k] = logf(u);
}

extern "C" __global__ void __launch_bounds__(SPEC_MAX_THREADS) bsvi_spec_kernel(const SpecArgs A_unused) {
    (void)A_unused;
    const SPEC_CONST_AS char* ka = (const SPEC_CONST_AS char*)__builtin_amdgcn_kernarg_segment_ptr();
#define SPEC_A ((const SPEC_CONST_AS SpecArgs*)ka)
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x, W = nthreads >> 6, wave = tid >> 6, lane = tid & 63u;
    const uint32_t G = gridDim.x;
    float* const WS = spec_lds + SPEC_OFF_WS;
    float* const RED = spec_lds + SPEC_OFF_RED;
    float* const PS = spec_lds + SPEC_OFF_PS;
    uint32_t* const TAB = reinterpret_cast<uint32_t*>(spec_lds + SPEC_OFF_TAB);
    float* const TR = spec_lds + SPEC_OFF_TR;
    float* const WSw = WS + wave * SPEC_NUG_PAD;
    float* const TRw = TR + wave * SPEC_TR_FLOATS;

    SpecLane T;
    T.lane = lane;
    T.n = blockIdx.x * nthreads + tid;
    {
        const uint32_t n_local = SPEC_A->n_local;
        T.active = T.n < n_local;
        T.nc = T.active ? T.n : (n_local - 1u);
        T.nidx = SPEC_A->sample_base + T.nc;
    }
    T.vz = T.n >> 31;
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
        for (uint32_t i = tid; i < 4u * SPEC_N_UNIFORM; i += nthreads) TAB[i] = uniform[i];
        for (uint32_t i = tid; i < SPEC_N_PARAMS + 1u; i += nthreads) TAB[SPEC_TAB_PTR + i] = pu_ptr[i];
        for (uint32_t j = tid; j < SPEC_N_UGRAD; j += nthreads) { TAB[SPEC_TAB_POS + j] = pu_pos[j]; TAB[SPEC_TAB_IDX + j] = pu_idx[j]; }
        for (uint32_t i = tid; i < SPEC_N_OBS; i += nthreads) spec_lds[SPEC_N_UNIFORM + i] = obs[i];
        const uint8_t* mask = SPEC_A->mask;
        const uint8_t* mask_first = SPEC_A->mask_first;
        for (uint32_t i = tid; i < SPEC_N_PARAMS; i += nthreads) {
            PS[i] = params[i];
            if (step) {
#pragma unroll
                for (uint32_t s = 0; s < 4u; ++s) PS[(1u + s) * SPEC_NP_PAD + i] = state[(size_t)s * SPEC_N_PARAMS + i];
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

    for (uint32_t it = 0; it < n_it; ++it) {
        // ---- one Monte-Carlo sample per lane.  Its standard normals do not depend on the uniform table: they are drawn
        //      ahead of the barrier that publishes it.
        {
            SPEC_RELOAD_ARGS();
            SpecBody B;
            B.n_local = SPEC_A->n_local;
            B.seed_lo = SPEC_A->seed_lo;
            B.seed_hi = SPEC_A->seed_hi;
#if SPEC_DIAG
            // a given-noise sequence is laid out [iteration][row][n_local]
            B.noise = SPEC_A->noise ? SPEC_A->noise + (size_t)it * SPEC_N_NOISE * B.n_local : nullptr;
            B.samples_out = SPEC_A->samples_out;
            B.noise_out = SPEC_A->noise_out;
            B.fvalue_out = SPEC_A->fvalue_out;
#else
            B.noise = nullptr; B.samples_out = nullptr; B.noise_out = nullptr; B.fvalue_out = nullptr;
#endif
            unsigned long long off = (((unsigned long long)SPEC_A->offset_hi << 32) | SPEC_A->offset_lo) + it;
            const unsigned long long* const offset_dev = SPEC_A->offset_dev;
            if (offset_dev) off += *offset_dev;
            T.f = 0.0f;
            T.lq = 0.0f;
            T.off_lo = (uint32_t)off;
            T.off_hi = (uint32_t)(off >> 32);
            SpecNoise Z;
            spec_draw(B, T, Z);
            __syncthreads();                                   // the uniform table of this iteration is complete
            spec_body(B, T, Z, TRw, WSw);
        }
        const float value = (SPEC_ESTIMATOR == BSVI_EST_BLACKBOX) ? (T.lq * T.f + T.f) : T.f;
        const float vsum = wave_sum(T.active ? value : 0.0f);
        const float bad = wave_sum((T.active && !isfinite(value)) ? 1.0f : 0.0f);
        if (lane == 0) { RED[8 + 2 * wave] = vsum; RED[9 + 2 * wave] = bad; }
        __syncthreads();                                       // every wave's sums are in WS / RED

        // ---- several workgroups: every one publishes its row of sums, the last to arrive adds the rows in order
        uint32_t rows = W;                                     // rows of WS / RED that hold sums
        if (G > 1) {
            SPEC_RELOAD_ARGS();
            float* const partials = SPEC_A->partials;
            unsigned int* const ticket = SPEC_A->ticket;
            const uint32_t stride = 2u + SPEC_N_UGRAD;
            float* mine = partials + (size_t)blockIdx.x * stride;
            for (uint32_t k = tid; k < SPEC_N_UGRAD; k += nthreads) {
                float s = WS[k];
                for (uint32_t w = 1; w < W; ++w) s += WS[w * SPEC_NUG_PAD + k];
                mine[2 + k] = s;
            }
            if (tid == 0) {
                float s = 0.0f, c = 0.0f;
                for (uint32_t w = 0; w < W; ++w) { s += RED[8 + 2 * w]; c += RED[9 + 2 * w]; }
                mine[0] = s;
                mine[1] = c;
            }
            __syncthreads();                                   // every wave's stores issued and drained
            if (tid == 0) {
                const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                RED[2] = (t == G - 1u) ? 1.0f : 0.0f;
            }
            __syncthreads();
            if (RED[2] == 0.0f) return;
            // slices of rows per column, then the slices in order (G can be thousands of workgroups)
            const uint32_t slices = (nthreads / stride) ? (nthreads / stride) : 1u;
            float* const SL = TR;                              // the transpose tiles are free now
            const uint32_t max_slices = (SPEC_MAX_WAVES * SPEC_TR_FLOATS) / stride;
            const uint32_t S = slices < max_slices ? slices : max_slices;
            for (uint32_t i = tid; i < S * stride; i += nthreads) {
                const uint32_t sl = i / stride, c = i - sl * stride;
                float s = 0.0f;
                for (uint32_t b = sl; b < G; b += S)
                    s += __hip_atomic_load(&partials[(size_t)b * stride + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                SL[i] = s;
            }
            __syncthreads();
            for (uint32_t c = tid; c < stride; c += nthreads) {
                float s = 0.0f;
                for (uint32_t sl = 0; sl < S; ++sl) s += SL[sl * stride + c];
                if (c < 2u) RED[8 + c] = s; else WS[c - 2u] = s;
            }
            if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            rows = 1;
        }

        // ---- epilogue.  The thread that owns parameter i does everything that depends on it: adds the waves' sums of
        //      its uniform entries, applies the chain rule, the optimizer step on the LDS working copy, and publishes
        //      the entries' new values for the next iteration — so an iteration of the in-kernel loop has two
        //      barriers: after the body, and before the next one.
        SPEC_RELOAD_ARGS();
        const uint32_t n_global = SPEC_A->n_global;
        float vs = 0.0f, vb = 0.0f;
        for (uint32_t w = 0; w < rows; ++w) { vs += RED[8 + 2 * w]; vb += RED[9 + 2 * w]; }
        const float loss = -vs / (float)n_global;
        const float finite = isfinite(loss) ? 1.0f : 0.0f;
        float* const out = SPEC_A->out;
        if (tid == 0) {
            out[0] = vs;
            out[1] = vb;
            if (step) {
                out[2] = loss;
                out[3] = finite;
                float* const loss_slot = SPEC_A->loss_slot;
                float* const finite_slot = SPEC_A->finite_slot;
                if (loss_slot) loss_slot[it] = loss;
                if (finite_slot) finite_slot[it] = finite;
            }
        }
        const float scale = step ? -1.0f / (float)n_global : 1.0f;
        const bool last = it + 1u == n_it;
        const uint32_t mask_bit = (mode == SPEC_MODE_LOOP && it <= SPEC_A->pretraining_iterations) ? 2u : 1u;
        for (uint32_t i = tid; i < SPEC_N_PARAMS; i += nthreads) {
            const float theta = PS[i];
            const uint32_t j0 = TAB[SPEC_TAB_PTR + i], j1 = TAB[SPEC_TAB_PTR + i + 1];
            float gsum = 0.0f;
            for (uint32_t j = j0; j < j1; ++j) {
                const uint32_t pos = TAB[SPEC_TAB_POS + j], k = TAB[SPEC_TAB_IDX + j];
                float s = WS[pos];
                for (uint32_t w = 1; w < rows; ++w) s += WS[w * SPEC_NUG_PAD + pos];
                gsum += s * (__uint_as_float(TAB[4 * k + 3]) * utransform_grad((int)(TAB[4 * k + 1] & 0xFFu), theta));
            }
            const float grad = gsum * scale;
            out[BSVI_OUT_HEADER + i] = grad;
            if (step && finite != 0.0f && (TAB[SPEC_TAB_MASK + i] & mask_bit)) {
                const bsvi_opt_cfg cfg = SPEC_A->cfg;
                optimizer_update(cfg, PS, PS + SPEC_NP_PAD, SPEC_NP_PAD, i, grad);
            }
            if (!step) continue;
            if (last) {
                float* const params = SPEC_A->params;
                float* const state = SPEC_A->state;
                params[i] = PS[i];
#pragma unroll
                for (uint32_t s = 0; s < 4u; ++s) state[(size_t)s * SPEC_N_PARAMS + i] = PS[(1u + s) * SPEC_NP_PAD + i];
            } else {
                const float theta2 = PS[i];
                for (uint32_t j = j0; j < j1; ++j) spec_publish_uniform(TAB, TAB[SPEC_TAB_IDX + j], theta2);
            }
        }
    }
#undef SPEC_A
}

}  // namespace bsvi
