// spec_main.h — second half of the frame around a program-specialised ELBO kernel (see spec_prelude.h): the kernel.
//
// One workgroup = up to SPEC_MAX_WAVES waves of 64 Monte-Carlo samples.  Per iteration:
//   prologue   U[k] = a + b*g(theta | const)                      (geometric_ranges.py transforms, once per workgroup)
//   body       spec_body(): GENERATED straight-line forward / reverse sweep of one sample per lane, registers only;
//              per-lane gradient contributions leave through the transpose tile (spec_du / spec_du_flush)
//   reduce     wave (DPP) -> workgroup (LDS) -> grid (partials + arrival ticket, last workgroup finishes); fixed
//              order everywhere, no float atomics: bitwise reproducible
//   epilogue   chain rule U -> theta through the CSR map; by mode: sums only (the caller all-reduces them over the
//              sample shards of other GPUs), or loss / finite flag / scaled gradients / fused SGD-Adam step
//              (inference.py:96-108), or that whole loop n_iterations times inside this one launch.
#pragma once

namespace bsvi {

// Scalar-register discipline (as in the interpreter's persistent kernels): SpecArgs is ~50 dwords.  Read as `A.x`
// they are all loaded at entry and stay live through the body, whose Philox key schedule and exec masks then spill
// through v_writelane / v_readlane (800 of 4 300 instructions at BASELINE config 1).  So every phase re-reads what it
// needs from the kernarg segment, through a pointer the optimiser cannot see through.
#if defined(__HIP_DEVICE_COMPILE__)
#define SPEC_CONST_AS __attribute__((address_space(4)))
#define SPEC_RELOAD_ARGS() asm volatile("" : "+s"(ka))
#else
#define SPEC_CONST_AS
#define SPEC_RELOAD_ARGS()
#endif

extern "C" __global__ void __launch_bounds__(SPEC_MAX_THREADS) bsvi_spec_kernel(const SpecArgs A_unused) {
    (void)A_unused;
    const SPEC_CONST_AS char* ka = (const SPEC_CONST_AS char*)__builtin_amdgcn_kernarg_segment_ptr();
#define SPEC_A ((const SPEC_CONST_AS SpecArgs*)ka)
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x, W = nthreads >> 6, wave = tid >> 6, lane = tid & 63u;
    const uint32_t G = gridDim.x;
    float* const WS = spec_lds + SPEC_U_PAD;
    float* const RED = WS + SPEC_MAX_WAVES * SPEC_NUG_PAD;
    float* const TR = RED + SPEC_RED_FLOATS;
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

    {
        const float* obs = SPEC_A->obs;
        for (uint32_t i = tid; i < SPEC_N_OBS; i += nthreads) spec_lds[SPEC_N_UNIFORM + i] = obs[i];
    }
    const uint32_t mode = SPEC_A->mode;
    const uint32_t n_it = (mode == SPEC_MODE_LOOP) ? SPEC_A->n_iterations : 1u;
    for (uint32_t it = 0; it < n_it; ++it) {
        // ---- prologue: the lane-uniform parameter transforms
        {
            SPEC_RELOAD_ARGS();
            const bsvi_uniform_entry* uniform = SPEC_A->uniform;
            const float* consts = SPEC_A->consts;
            const float* params = SPEC_A->params;
            for (uint32_t k = tid; k < SPEC_N_UNIFORM; k += nthreads) {
                const bsvi_uniform_entry e = uniform[k];
                // agent-scope load: in loop mode the previous iteration's optimizer step rewrote the parameters
                const float x = e.is_param ? __hip_atomic_load(&params[e.src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : consts[e.src];
                spec_lds[k] = e.a + e.b * utransform(e.transform, x);
            }
        }
        __syncthreads();

        // ---- one Monte-Carlo sample per lane
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
            const uint32_t off_lo = SPEC_A->offset_lo;
            T.f = 0.0f;
            T.lq = 0.0f;
            T.off_lo = off_lo + it;
            T.off_hi = SPEC_A->offset_hi + ((T.off_lo < off_lo) ? 1u : 0u);
            spec_body(B, T, TRw, WSw);
        }
        const float value = (SPEC_ESTIMATOR == BSVI_EST_BLACKBOX) ? (T.lq * T.f + T.f) : T.f;
        const float vsum = wave_sum(T.active ? value : 0.0f);
        const float bad = wave_sum((T.active && !isfinite(value)) ? 1.0f : 0.0f);
        if (lane == 0) { RED[8 + 2 * wave] = vsum; RED[9 + 2 * wave] = bad; }
        __syncthreads();

        // ---- workgroup totals, waves in order
        for (uint32_t k = tid; k < SPEC_N_UGRAD; k += nthreads) {
            float s = WS[k];
            for (uint32_t w = 1; w < W; ++w) s += WS[w * SPEC_NUG_PAD + k];
            WS[k] = s;
        }
        if (tid == 0) {
            float s = 0.0f, c = 0.0f;
            for (uint32_t w = 0; w < W; ++w) { s += RED[8 + 2 * w]; c += RED[9 + 2 * w]; }
            RED[0] = s;
            RED[1] = c;
        }
        __syncthreads();

        // ---- grid totals: every workgroup publishes its row, the last one to arrive adds the rows in order
        if (G > 1) {
            SPEC_RELOAD_ARGS();
            float* const partials = SPEC_A->partials;
            unsigned int* const ticket = SPEC_A->ticket;
            const uint32_t stride = 2u + SPEC_N_UGRAD;
            float* mine = partials + (size_t)blockIdx.x * stride;
            for (uint32_t k = tid; k < SPEC_N_UGRAD; k += nthreads) mine[2 + k] = WS[k];
            if (tid == 0) { mine[0] = RED[0]; mine[1] = RED[1]; }
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
                if (c < 2u) RED[c] = s; else WS[c - 2u] = s;
            }
            if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }

        // ---- epilogue
        SPEC_RELOAD_ARGS();
        const bool step = mode != SPEC_MODE_SUMS;
        const uint32_t n_global = SPEC_A->n_global, n_params = SPEC_A->n_params;
        const float loss = -RED[0] / (float)n_global;
        const float finite = isfinite(loss) ? 1.0f : 0.0f;
        float* const out = SPEC_A->out;
        if (tid == 0) {
            out[0] = RED[0];
            out[1] = RED[1];
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
        const uint8_t* const mask = (mode == SPEC_MODE_LOOP && it <= SPEC_A->pretraining_iterations) ? SPEC_A->mask_first : SPEC_A->mask;
        float* const params = SPEC_A->params;
        float* const state = SPEC_A->state;
        const uint32_t* const pu_ptr = SPEC_A->pu_ptr;
        const uint32_t* const pu_pos = SPEC_A->pu_pos;
        const uint32_t* const pu_idx = SPEC_A->pu_idx;
        const bsvi_uniform_entry* const uniform = SPEC_A->uniform;
        for (uint32_t i = tid; i < n_params; i += nthreads) {
            float gsum = 0.0f;
            const float theta = __hip_atomic_load(&params[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t j = pu_ptr[i]; j < pu_ptr[i + 1]; ++j) {
                const bsvi_uniform_entry e = uniform[pu_idx[j]];
                gsum += WS[pu_pos[j]] * (e.b * utransform_grad(e.transform, theta));
            }
            const float grad = gsum * scale;
            out[BSVI_OUT_HEADER + i] = grad;
            if (step && finite != 0.0f && mask[i]) {
                const bsvi_opt_cfg cfg = SPEC_A->cfg;
                optimizer_update(cfg, params, state, n_params, i, grad);
            }
        }
        // loop mode: the parameter stores are drained by this barrier; the next prologue re-reads them past the L1
        __syncthreads();
    }
#undef SPEC_A
}

}  // namespace bsvi
