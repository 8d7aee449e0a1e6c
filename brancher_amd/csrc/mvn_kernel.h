// mvn_kernel.h — frame of the batched multivariate-normal kernel ("batched Cholesky / triangular-solve
// kernels"): log N(x | m, C) of `brancher/distributions.py:314-331` (torch MultivariateNormal(covariance_matrix=C)) and its
// gradient, for a covariance that is an ELEMENTWISE link expression C_ij = g(M1_ij, M2_ij, ...; s_1 .. s_m) of constant
// matrices and a few per-sample / learnable scalars — a Gaussian process whose kernel hyper-parameters are inferred
// (`stochastic_processes.py:29-40`, `standard_variables.py:317-347`).  mvn.cpp generates `mvn_cov` (the expression with its
// forward-mode derivatives in the m scalars) in front of this header and hiprtc compiles the two together.
//
// ONE WAVE per Monte-Carlo sample, everything in LDS, no barrier between dependent steps (a wave's LDS operations complete
// in program order; __syncthreads() of a one-wave workgroup is a wait + a free barrier and only fences the compiler):
//   1  C (lower triangle) from mvn_cov                          A <- C
//   2  Cholesky, left-looking, in place                         A <- L          (sum log L_ii on the way)
//   3  X = L^-1, column j by lane j, stored transposed          XT[j][k] = X[k][j]
//   4  S = C^-1 = X^T X, row i by lane i                        A <- S          (both triangles)
//   5  alpha = S d,  quad = d.alpha,  log p = -quad/2 - sum log L_ii - D/2 log 2pi
//   6  dlogp/dC = (alpha alpha^T - S)/2  contracted with the expression's derivatives -> dlogp/ds_k;  dlogp/dx = -alpha
// The other two parameterisations of `distributions.py:314-331` run through the same steps with the roles changed (MVN_FORM):
//   scale_tril        the expression IS L (its lower triangle): step 2 only adds up log L_ii; with y = L^-1 d (= X d)
//                     d log p / d L_ij = alpha_i y_j - [i = j] / L_ii   for i >= j  (alpha = L^-T y = C^-1 d as before)
//   precision_matrix  alpha = P d and the quadratic form come first (from the expression's lower triangle); then P = M M^T is
//                     factorised where C was, log p = -d.alpha / 2 + sum log M_ii - D/2 log 2pi, S = X^T X with X = M^-1 is
//                     P^-1, and  d log p / d P = (S - d d^T) / 2
// Rows are 16-byte aligned with a stride of 4 * odd words: every inner product runs on ds_read_b128 along k, conflict-free
// across the lanes' rows; both matrices start zeroed so that aligned 4-wide blocks may overrun a triangle's edge.
//
// Rows out: coefficients of the slot inputs | of x (when latent) | of the uniform inputs | of m (when learnable) | e.
// The results leave as the rows of a LINEAR surrogate (lowering.ExternalMvn): g_k = dlogp/d(input k) and
// e = logp - sum_k g_k input_k, so that  e + sum_k g_k input_k  has the value AND the gradient of log p at this sample —
// the scalar program adds it to f through BSVI_DIST_LINEAR terms and its reverse sweep carries g_k on.
#pragma once

namespace bsvi {

struct MvnArgs {
    const float* samples;                 // [rows][n_local]: slot values of the draw (samples_out of the base program)
    const float* params;
    const float* mats;                    // [MVN_NMATS][D][D]
    const float* vecs;                    // [2][D]: loc, observed value
    const bsvi_uniform_entry* uniform_inputs;
    const bsvi_uniform_entry* loc_entries;    // [D] when the loc is learnable (MVN_LOC_PARAM), else null
    float* rows_out;                      // the surrogate's rows: [n_rows_out][n_local], row 0 = first input's coefficient
    uint32_t n_local, value_row0;
    uint32_t input_rows[8];
    float weight;
    uint32_t reserved;
};

#ifndef MVN_NIN_PAD
#define MVN_NIN_PAD (MVN_NIN > 0 ? MVN_NIN : 1)
#endif

typedef float mvn_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float mvn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float mvn_dot4(mvn_f4 a, mvn_f4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

extern "C" __global__ void __launch_bounds__(64) bsvi_mvn_kernel(const MvnArgs G) {
    constexpr int D = MVN_D, LD = MVN_LD;
    __shared__ __attribute__((aligned(16))) float A[D * LD];
    __shared__ __attribute__((aligned(16))) float XT[D * LD];
    __shared__ __attribute__((aligned(16))) float dvec[(D + 3) / 4 * 4 + 4];
    __shared__ __attribute__((aligned(16))) float avec[(D + 3) / 4 * 4 + 4];
    __shared__ float inputs[MVN_NIN_PAD];
    const int lane = threadIdx.x;
    const uint32_t n = blockIdx.x;
    if (n >= G.n_local) return;

    // ---- inputs of the covariance expression, d = x - m, zeroed matrices
    if (lane < MVN_NIN) {
        float v;
        if (lane < MVN_NSI) {
            v = G.samples[(size_t)G.input_rows[lane] * G.n_local + n];
        } else {
            const bsvi_uniform_entry e = G.uniform_inputs[lane - MVN_NSI];
            v = e.a + e.b * utransform(e.transform, G.params[e.src]);
        }
        inputs[lane] = v;
    }
    for (int i = lane; i < (D + 3) / 4 * 4 + 4; i += 64) {
        float v = 0.0f;
        if (i < D) {
            const float x = MVN_VALUE_LATENT ? G.samples[(size_t)(G.value_row0 + i) * G.n_local + n] : G.vecs[D + i];
            float m = G.vecs[i];
            if (MVN_LOC_PARAM) {
                const bsvi_uniform_entry e = G.loc_entries[i];
                m = e.a + e.b * utransform(e.transform, G.params[e.src]);
            }
            v = x - m;
        }
        dvec[i] = v;
        avec[i] = 0.0f;
    }
    for (int i = lane; i < D * LD / 4; i += 64) {
        reinterpret_cast<mvn_f4*>(A)[i] = mvn_f4{0.0f, 0.0f, 0.0f, 0.0f};
        reinterpret_cast<mvn_f4*>(XT)[i] = mvn_f4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    __syncthreads();
    float in[MVN_NIN_PAD];
#pragma unroll
    for (int k = 0; k < MVN_NIN_PAD; ++k) in[k] = k < MVN_NIN ? inputs[k] : 0.0f;

    // ---- 1: the lower triangle of C
    constexpr int NTRI = D * (D + 1) / 2;
    for (int e = lane; e < NTRI; e += 64) {
        int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while (i * (i + 1) / 2 > e) --i;
        while ((i + 1) * (i + 2) / 2 <= e) ++i;
        const int j = e - i * (i + 1) / 2;
        float c, dc[MVN_NIN_PAD];
        mvn_cov(i, j, in, G.mats, c, dc);
        A[i * LD + j] = c;
    }
    __syncthreads();

#if MVN_FORM == 2
    // ---- precision form: alpha = P d from the lower triangle (row part along k, column part down the rows), before P is factorised
    float quad = 0.0f;
    for (int t = 0; t < (D + 63) / 64; ++t) {
        const int i = lane + 64 * t;
        if (i < D) {
            float acc = 0.0f;
            for (int k = 0; k <= i; ++k) acc += A[i * LD + k] * dvec[k];
            for (int k = i + 1; k < D; ++k) acc += A[k * LD + i] * dvec[k];
            avec[i] = acc;
            quad += acc * dvec[i];
        }
    }
    quad = mvn_wave_sum(quad);
    __syncthreads();
#endif

    // ---- 2: Cholesky (left-looking): column j from the columns before it
    float logdet = 0.0f;
#if MVN_FORM == 1
    // (scale_tril: the lower triangle already holds L)
    for (int t = 0; t < (D + 63) / 64; ++t) {
        const int i = lane + 64 * t;
        if (i < D) logdet += logf(A[i * LD + i]);
    }
    logdet = mvn_wave_sum(logdet);
#else
    for (int j = 0; j < D; ++j) {
        float s[(D + 63) / 64];
        const int jb = j & ~3;
#pragma unroll
        for (int t = 0; t < (D + 63) / 64; ++t) {
            const int i = j + lane + 64 * t;
            float acc = 0.0f;
            if (i < D) {
                const mvn_f4* ri = reinterpret_cast<const mvn_f4*>(A + i * LD);
                const mvn_f4* rj = reinterpret_cast<const mvn_f4*>(A + j * LD);
                for (int k = 0; k < jb; k += 4) acc += mvn_dot4(ri[k >> 2], rj[k >> 2]);
                // the block that holds column j itself: only the components in front of it
                const mvn_f4 a = ri[jb >> 2], b = rj[jb >> 2];
                const int r = j - jb;
                acc += (r > 0 ? a.x * b.x : 0.0f) + (r > 1 ? a.y * b.y : 0.0f) + (r > 2 ? a.z * b.z : 0.0f);
                acc = A[i * LD + j] - acc;
            }
            s[t] = acc;
        }
        const float pivot = __shfl(s[0], 0, 64);          // row j is lane 0 of the first pass
        const float ljj = sqrtf(pivot), inv = 1.0f / ljj;  // (a pivot <= 0 gives NaN: the step is then skipped as non-finite)
        logdet += logf(ljj);
#pragma unroll
        for (int t = 0; t < (D + 63) / 64; ++t) {
            const int i = j + lane + 64 * t;
            if (i < D) A[i * LD + j] = (i == j) ? ljj : s[t] * inv;
        }
        __syncthreads();
    }
#endif

    // ---- 3: X = L^-1, column j by lane j, kept transposed: XT[j][i] = X[i][j] = -(sum_{j<=k<i} L[i][k] X[k][j]) / L[i][i]
    for (int t = 0; t < (D + 63) / 64; ++t) {
        const int j = lane + 64 * t;
        if (j < D) {
            float* xj = XT + j * LD;
            xj[j] = 1.0f / A[j * LD + j];
            const int jb = j & ~3;
            for (int i = j + 1; i < D; ++i) {
                const mvn_f4* li = reinterpret_cast<const mvn_f4*>(A + i * LD);
                const mvn_f4* xr = reinterpret_cast<const mvn_f4*>(xj);
                float acc = 0.0f;
                // (aligned blocks: XT[j][k < j] and XT[j][k >= i] are still zero, A above its diagonal is zero)
                for (int k = jb; k < i; k += 4) acc += mvn_dot4(li[k >> 2], xr[k >> 2]);
                xj[i] = -acc / A[i * LD + i];
            }
        }
    }
    __syncthreads();

#if MVN_FORM == 1
    // ---- scale_tril form: y = L^-1 d = X d, y_j = sum_{k <= j} X[j][k] d_k = sum_k XT[k][j] d_k (lanes along j: conflict-free)
    __shared__ __attribute__((aligned(16))) float yvec[(D + 3) / 4 * 4 + 4];
    for (int t = 0; t < (D + 63) / 64; ++t) {
        const int j = lane + 64 * t;
        if (j < D) {
            float acc = 0.0f;
            for (int k = 0; k <= j; ++k) acc += XT[k * LD + j] * dvec[k];
            yvec[j] = acc;
        }
    }
    __syncthreads();
#endif

    // ---- 4: S = X^T X:  S[i][j] = sum_{k >= i} XT[i][k] XT[j][k]  (i >= j), written to both triangles of A
    for (int t = 0; t < (D + 63) / 64; ++t) {
        const int i = lane + 64 * t;
        if (i < D) {
            const int ib = i & ~3;
            const mvn_f4* xi = reinterpret_cast<const mvn_f4*>(XT + i * LD);
            for (int j = 0; j <= i; ++j) {
                const mvn_f4* xj = reinterpret_cast<const mvn_f4*>(XT + j * LD);
                float acc = 0.0f;
                for (int k = ib; k < D; k += 4) acc += mvn_dot4(xi[k >> 2], xj[k >> 2]);      // (XT[i][k < i] = 0; columns >= D are zero)
                A[i * LD + j] = acc;
                A[j * LD + i] = acc;
            }
        }
    }
    __syncthreads();

    // ---- 5: alpha = S d, the quadratic form, log p
#if MVN_FORM != 2
    float quad = 0.0f;
    for (int t = 0; t < (D + 63) / 64; ++t) {
        const int i = lane + 64 * t;
        if (i < D) {
            const mvn_f4* si = reinterpret_cast<const mvn_f4*>(A + i * LD);
            const mvn_f4* dv = reinterpret_cast<const mvn_f4*>(dvec);
            float acc = 0.0f;
            for (int k = 0; k < D; k += 4) {
                mvn_f4 row = si[k >> 2];
                // (the row's tail beyond column D-1 may hold nothing of S: d is zero there)
                acc += mvn_dot4(row, dv[k >> 2]);
            }
            avec[i] = acc;
            quad += acc * dvec[i];
        }
    }
    quad = mvn_wave_sum(quad);
    __syncthreads();
    const float logp = -0.5f * quad - logdet - 0.5f * (float)D * 1.8378770664093453f;
#else
    const float logp = -0.5f * quad + logdet - 0.5f * (float)D * 1.8378770664093453f;      // (logdet = sum log M_ii = log det P / 2)
#endif

    // ---- 6: dlogp/ds_k = sum_{i >= j} m_ij (alpha_i alpha_j - S_ij) / 2 * dC_ij/ds_k   (m_ij = 2 off the diagonal: C is symmetric)
    float gin[MVN_NIN_PAD];
#pragma unroll
    for (int k = 0; k < MVN_NIN_PAD; ++k) gin[k] = 0.0f;
#if MVN_NIN > 0
    for (int e = lane; e < NTRI; e += 64) {
        int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while (i * (i + 1) / 2 > e) --i;
        while ((i + 1) * (i + 2) / 2 <= e) ++i;
        const int j = e - i * (i + 1) / 2;
        float c, dc[MVN_NIN_PAD];
        mvn_cov(i, j, in, G.mats, c, dc);
#if MVN_FORM == 0
        const float gij = (i == j ? 0.5f : 1.0f) * (avec[i] * avec[j] - A[i * LD + j]);
#elif MVN_FORM == 1
        const float gij = avec[i] * yvec[j] - (i == j ? XT[i * LD + i] : 0.0f);           // (X_ii = 1 / L_ii)
#else
        const float gij = (i == j ? 0.5f : 1.0f) * (A[i * LD + j] - dvec[i] * dvec[j]);
#endif
#pragma unroll
        for (int k = 0; k < MVN_NIN; ++k) gin[k] += gij * dc[k];
    }
#pragma unroll
    for (int k = 0; k < MVN_NIN; ++k) gin[k] = mvn_wave_sum(gin[k]);
#endif

    // ---- the surrogate's rows: coefficients of the slot inputs, of x (when latent), of the uniform inputs, then e
    const float w = G.weight;
    float lin = 0.0f;              // sum_k g_k * input_k over everything that carries a coefficient
#pragma unroll
    for (int k = 0; k < MVN_NIN; ++k) lin += w * gin[k] * in[k];
    float linx = 0.0f;
    uint32_t row = 0;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < MVN_NSI; ++k) G.rows_out[(size_t)(row + k) * G.n_local + n] = w * gin[k];
    }
    row += MVN_NSI;
    if (MVN_VALUE_LATENT) {
        for (int t = 0; t < (D + 63) / 64; ++t) {
            const int i = lane + 64 * t;
            if (i < D) {
                const float gx = -w * avec[i];
                G.rows_out[(size_t)(row + i) * G.n_local + n] = gx;
                linx += gx * G.samples[(size_t)(G.value_row0 + i) * G.n_local + n];
            }
        }
        linx = mvn_wave_sum(linx);
        row += D;
    }
    if (lane == 0) {
#pragma unroll
        for (int k = MVN_NSI; k < MVN_NIN; ++k) G.rows_out[(size_t)(row + k - MVN_NSI) * G.n_local + n] = w * gin[k];
    }
    row += MVN_NIN - MVN_NSI;
    float linm = 0.0f;
    if (MVN_LOC_PARAM) {           // d log p / d m = +alpha
        for (int t = 0; t < (D + 63) / 64; ++t) {
            const int i = lane + 64 * t;
            if (i < D) {
                const float gm = w * avec[i];
                G.rows_out[(size_t)(row + i) * G.n_local + n] = gm;
                const bsvi_uniform_entry e = G.loc_entries[i];
                linm += gm * (e.a + e.b * utransform(e.transform, G.params[e.src]));
            }
        }
        linm = mvn_wave_sum(linm);
        row += D;
    }
    if (lane == 0) G.rows_out[(size_t)row * G.n_local + n] = w * logp - lin - linx - linm;
}

}  // namespace bsvi
