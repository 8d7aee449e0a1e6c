// reduce_kernel.h — frame of the REDUCE node (round 6): a link with a reduction over more elements than the per-sample program unrolls,
//     r[n][d] = sum_e A(e; s_n) * X[d][e],
// `mean_response = BF.sum(BF.sum(receptive_field * input, dim=1), dim=2)` of examples/PopulationReceptiveFields.py:29-31 — A an
// ELEMENTWISE link expression of constant matrices (the x / y meshes) and a few scalars that differ per Monte-Carlo sample (mu_x, mu_y,
// v) or are learnable, X a data matrix [datapoints][elements] that is new in every evaluation (the example's `input` node, observed by
// flag only: the reference draws it once per evaluation, variables.py:553-565).  mvn.cpp generates `red_fn` (the expression with its
// forward-mode derivatives in the scalars — the generator of the batched multivariate-normal kernel) in front of this header.
//
// ONE WORKGROUP per Monte-Carlo sample: (1) A and dA/ds_k at every element into LDS, [(K + 1)][E]; (2) every (k, d) pair is ONE wave's dot
// product with row d of X (lanes along e: coalesced reads of X, which the L2 holds — it is shared by every sample), a fixed butterfly;
// (3) the rows of the LINEAR surrogate  r_d = e_d + sum_k g_dk s_k  (value and gradient of r_d at the sample, as the multivariate-normal
// node's): rows k * D + d = g_dk, rows K * D + d = e_d = r_d - sum_k g_dk s_k, and one last row: the log-probability of the drawn data
// node (the same for every sample; lowering.reduce_external adds it to f).  The per-sample program composes r_d from these GIVEN rows.
#pragma once

namespace bsvi {

struct ReduceArgs {
    const float* samples;                 // [rows][n_local]: slot values of the draw (samples_out of the base program)
    const float* params;
    const float* mats;                    // [RED_NMATS][RED_I][RED_J]
    const float* data;                    // X [n_data][RED_E]
    const bsvi_uniform_entry* uniform_inputs;
    float* rows_out;                      // [(RED_NIN + 1) * n_data + 1][n_local]
    const float* logp;                    // [1]: weight * log p of the drawn data node, or null (0)
    uint32_t n_local, n_data;
    uint32_t input_rows[8];
};

#ifndef RED_NIN_PAD
#define RED_NIN_PAD (RED_NIN > 0 ? RED_NIN : 1)
#endif

extern "C" __global__ void __launch_bounds__(256) bsvi_reduce_kernel(const ReduceArgs A) {
    extern __shared__ float red_lds[];
    float* const V = red_lds;                                   // [(RED_NIN + 1)][RED_E]
    float* const R = red_lds + (size_t)(RED_NIN + 1) * RED_E;  // [(RED_NIN + 1)][n_data]
    const uint32_t n = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    float in[RED_NIN_PAD];
#pragma unroll
    for (uint32_t k = 0; k < RED_NIN_PAD; ++k) in[k] = 0.0f;
#pragma unroll
    for (uint32_t k = 0; k < RED_NSI; ++k) in[k] = A.samples[(size_t)A.input_rows[k] * A.n_local + n];
#pragma unroll
    for (uint32_t k = RED_NSI; k < RED_NIN; ++k) {
        const bsvi_uniform_entry u = A.uniform_inputs[k - RED_NSI];
        in[k] = u.a + u.b * utransform(u.transform, A.params[u.src]);
    }
    for (uint32_t e = tid; e < RED_E; e += 256u) {
        float c, dc[RED_NIN_PAD];
        red_fn((int)(e / RED_J), (int)(e % RED_J), in, A.mats, c, dc);
        V[e] = c;
#pragma unroll
        for (uint32_t k = 0; k < RED_NIN; ++k) V[(size_t)(1u + k) * RED_E + e] = dc[k];
    }
    __syncthreads();
    const uint32_t pairs = (RED_NIN + 1u) * A.n_data;
    for (uint32_t p = wave; p < pairs; p += 4u) {
        const uint32_t k = p / A.n_data, d = p - k * A.n_data;
        const float* const v = V + (size_t)k * RED_E;
        const float* const x = A.data + (size_t)d * RED_E;
        float s = 0.0f;
        for (uint32_t e = lane; e < RED_E; e += 64u) s += v[e] * x[e];
        s = wave_sum(s);
        if (lane == 0u) R[p] = s;
    }
    __syncthreads();
    for (uint32_t d = tid; d < A.n_data; d += 256u) {
        float ed = R[d];
#pragma unroll
        for (uint32_t k = 0; k < RED_NIN; ++k) {
            const float g = R[(size_t)(1u + k) * A.n_data + d];
            A.rows_out[((size_t)k * A.n_data + d) * A.n_local + n] = g;
            ed -= g * in[k];
        }
        A.rows_out[((size_t)RED_NIN * A.n_data + d) * A.n_local + n] = ed;
    }
    if (tid == 0u) A.rows_out[(size_t)(RED_NIN + 1u) * A.n_data * A.n_local + n] = A.logp ? A.logp[0] : 0.0f;
}

}  // namespace bsvi
