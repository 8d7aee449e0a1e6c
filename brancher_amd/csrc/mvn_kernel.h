// mvn_kernel.h — frame of the batched multivariate-normal kernel ("batched Cholesky / triangular-solve
// kernels"): log N(x | m, C) of `brancher/distributions.py:314-331` (torch MultivariateNormal(covariance_matrix=C)) and its
// gradient, for a covariance that is an ELEMENTWISE link expression C_ij = g(M1_ij, M2_ij, ...; s_1 .. s_m) of constant
// matrices and a few per-sample / learnable scalars — a Gaussian process whose kernel hyper-parameters are inferred
// (`stochastic_processes.py:29-40`, `standard_variables.py:317-347`).  mvn.cpp generates `mvn_cov` (the expression with its
// forward-mode derivatives in the m scalars) in front of this header and hiprtc compiles the two together.
//
// ONE WORKGROUP of four waves per Monte-Carlo sample, ONE D x LD matrix in LDS that L, X^T and S share (round 4; round 3: one wave,
// two matrices — profiles/r4/mvn_steps.txt has the step times of every stage: D = 128 1179 -> 158 us per launch at 512 samples):
//   1  C (lower triangle) from mvn_cov, dealt out over the 256 threads             A(lower) <- C
//   2  Cholesky, left-looking, in PANELS of four columns: a row belongs to a pair    A(lower) <- L, diag / rdiag <- L_ii, 1 / L_ii
//      of lanes that split its 16-byte blocks; the 4 x 4 diagonal block goes through
//      LDS and every thread factorises it in registers: two barriers per panel
//   3  X = L^-1, column j by lane pair j in ROW BLOCKS of four, stored transposed    A(upper)[j][i] <- X[i][j], A[j][j] <- 1 / L_jj
//      in the triangle L does not use
//   4  S = C^-1 = X^T X, its D (D + 1) / 2 elements dealt out over the threads       A(strictly lower) <- S, sdiag <- S_ii
//   5  alpha = S d,  quad = d.alpha,  log p = -quad/2 - sum log L_ii - D/2 log 2pi
//   6  dlogp/dC = (alpha alpha^T - S)/2  contracted with the expression's derivatives -> dlogp/ds_k;  dlogp/dx = -alpha
// D is padded to whole blocks of four rows with an identity block (it factorises to itself and contributes nothing).
// The other two parameterisations of `distributions.py:314-331` run through the same steps with the roles changed (MVN_FORM):
//   scale_tril        the expression IS L (its lower triangle): step 2 only adds up log L_ii; with y = L^-1 d (= X d)
//                     d log p / d L_ij = alpha_i y_j - [i = j] / L_ii   for i >= j  (alpha = L^-T y = C^-1 d as before)
//   precision_matrix  alpha = P d and the quadratic form come first (from the expression's lower triangle); then P = M M^T is
//                     factorised where C was, log p = -d.alpha / 2 + sum log M_ii - D/2 log 2pi, S = X^T X with X = M^-1 is
//                     P^-1, and  d log p / d P = (S - d d^T) / 2
// Rows are 16-byte aligned with a stride of 4 * odd words: every inner product runs on ds_read_b128 along k; the matrix starts
// zeroed, and a reader of a row that holds two triangles masks the components of the other one.
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
    const bsvi_uniform_entry* value_entries;  // [D] when the value is learnable parameters (MVN_VALUE_PARAM: the taylor1 program), else null
    float* rows_out;                      // the surrogate's rows: [n_rows_out][n_local], row 0 = first input's coefficient
    float* scratch;                       // MVN_SPILL: [n_local][DP * LD] — the matrix of a sample when it does not fit LDS, else null
    uint32_t n_local, value_row0;
    uint32_t input_rows[8];
    float weight;
    uint32_t reserved;
};

#ifndef MVN_VALUE_PARAM
#define MVN_VALUE_PARAM 0
#endif
#ifndef MVN_NIN_PAD
#define MVN_NIN_PAD (MVN_NIN > 0 ? MVN_NIN : 1)
#endif
// MVN_SPILL (D > 192: DP x LD floats no longer fit the CU's 160 KiB): the matrix of a sample lives in its own DP x LD block of
// device memory instead — the same steps on the same layout through the CU's vector cache and the L2.  A workgroup's waves share
// one vector L1 and __syncthreads() waits for the stores, so what a step wrote is what the next step reads; the vectors and the
// panel's 4 x 4 block stay in LDS.
#ifndef MVN_SPILL
#define MVN_SPILL 0
#endif

typedef float mvn_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float mvn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float mvn_dot4(mvn_f4 a, mvn_f4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// timing experiments (BSVI_SPEC_DEFINES="#define MVN_STOP_AFTER k", tools/r4/mvn_steps.sh): the kernel ends behind step k
#if defined(MVN_STOP_AFTER)
#define MVN_STEP_END(k) if (MVN_STOP_AFTER == (k)) { if (tid == 0) G.rows_out[n] = A[0] + A[1] + avec[0]; return; }
#else
#define MVN_STEP_END(k)
#endif

constexpr int MVN_THREADS = 256;

// sum over the workgroup's four waves; `red` is free again when it returns
__device__ __forceinline__ float mvn_block_sum(float v, float* red, int tid) {
    v = mvn_wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}
// (i, j) of the e-th element of the lower triangle, row by row
__device__ __forceinline__ void mvn_tri(int e, int& i, int& j) {
    i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    while (i * (i + 1) / 2 > e) --i;
    while ((i + 1) * (i + 2) / 2 <= e) ++i;
    j = e - i * (i + 1) / 2;
}
// sum over the 16-byte blocks b0, b0 + step, ... < b1 of two LDS rows, four blocks in flight (one wave per SIMD has nobody to
// hide an LDS round trip behind but its own other loads: with one block per trip these loops ran at a tenth of the LDS rate).
// (Measured and dropped: eight blocks per trip with the tail replaced by zeros instead of a remainder loop — the selects and
// index arithmetic of the predication cost more issue slots than the round trips they save: D = 128 444 -> 514 us.)
__device__ __forceinline__ float mvn_rowdot(const mvn_f4* x, const mvn_f4* y, int b0, int b1, int step) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int b = b0;
    for (; b + 3 * step < b1; b += 4 * step) {
        const mvn_f4 x0 = x[b], x1 = x[b + step], x2 = x[b + 2 * step], x3 = x[b + 3 * step];
        const mvn_f4 y0 = y[b], y1 = y[b + step], y2 = y[b + 2 * step], y3 = y[b + 3 * step];
        a0 += mvn_dot4(x0, y0); a1 += mvn_dot4(x1, y1); a2 += mvn_dot4(x2, y2); a3 += mvn_dot4(x3, y3);
    }
    for (; b < b1; b += step) a0 += mvn_dot4(x[b], y[b]);
    return (a0 + a1) + (a2 + a3);
}

// the other lane of the pair (lane ^ 1): one DPP move, no LDS round trip (ds_bpermute, which __shfl_xor compiles to, is one)
__device__ __forceinline__ float mvn_pair(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));      // quad_perm [1, 0, 3, 2]
#else
    return v;
#endif
}
// x.y and y.y over the same blocks in ONE loop (the pivot of a Cholesky column rides on the loads of the row's own product)
__device__ __forceinline__ void mvn_rowdot2(const mvn_f4* x, const mvn_f4* y, int b0, int b1, int step, float& xy, float& yy) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
    int b = b0;
    for (; b + 3 * step < b1; b += 4 * step) {
        const mvn_f4 x0 = x[b], x1 = x[b + step], x2 = x[b + 2 * step], x3 = x[b + 3 * step];
        const mvn_f4 y0 = y[b], y1 = y[b + step], y2 = y[b + 2 * step], y3 = y[b + 3 * step];
        a0 += mvn_dot4(x0, y0); a1 += mvn_dot4(x1, y1); a2 += mvn_dot4(x2, y2); a3 += mvn_dot4(x3, y3);
        p0 += mvn_dot4(y0, y0); p1 += mvn_dot4(y1, y1); p2 += mvn_dot4(y2, y2); p3 += mvn_dot4(y3, y3);
    }
    for (; b < b1; b += step) { const mvn_f4 x0 = x[b], y0 = y[b]; a0 += mvn_dot4(x0, y0); p0 += mvn_dot4(y0, y0); }
    xy = (a0 + a1) + (a2 + a3);
    yy = (p0 + p1) + (p2 + p3);
}
// 1 / sqrt(p): the hardware estimate and one Newton step (the IEEE sqrt and divide of the library are ~40 dependent instructions
// in the middle of the one chain a Cholesky column is); NaN for p < 0 like sqrtf, so a non-positive pivot still poisons the step
__device__ __forceinline__ float mvn_rsqrt(float p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rsqf(p);
#else
    const float r = 1.0f / sqrtf(p);
#endif
    return p > 0.0f ? r * (1.5f - 0.5f * p * r * r) : __int_as_float(0x7fc00000);
}

// Round 4: FOUR waves per sample (round 3: one; tools/r4/mvn_steps.sh has the step times of both).  A row of the factorisation and
// a column of the triangular inverse belong to a PAIR of neighbouring lanes that split the inner product's 16-byte blocks
// between them (even / odd) and add their halves with one DPP exchange; the elements of C, of S = X^T X and of the gradient
// contraction are dealt out over all 256 threads.  One barrier per Cholesky column: every thread forms the pivot itself, from
// the blocks of row j it reads anyway, in the same order as everybody else — no broadcast of the pivot through LDS.
extern "C" __global__ void __launch_bounds__(MVN_THREADS) bsvi_mvn_kernel(const MvnArgs G) {
    // DP: D rounded up to whole 16-byte blocks of rows and columns.  The pad is an IDENTITY block (C_kk = 1 behind row D - 1): it
    // factorises to itself, contributes log 1 to the determinant and nothing to alpha or S, and lets every panel of four
    // columns below be a whole one.
    constexpr int D = MVN_D, LD = MVN_LD, DP = (D + 3) / 4 * 4, VP = DP + 4;
    // ONE matrix (round 3 and the first version of this kernel kept two: 136 x 136 at most, and one sample per CU from D = 100):
    //   below the diagonal   C, then L, at the end S = C^-1 (strictly lower part; its diagonal in sdiag)
    //   above the diagonal   zero until step 3, then X^T:  A[j][i] = X[i][j] for i > j   (X = L^-1)
    //   the diagonal         C_jj, L_jj, from step 3 on X_jj = 1 / L_jj
    // The diagonals of L live in diag / rdiag.  Readers of a row that mixes the two (the first block of a column of X, the
    // first block of a row in S = X^T X) mask the components that belong to the other triangle.
#if MVN_SPILL
    float* const A = G.scratch + (size_t)blockIdx.x * (size_t)(DP * LD);
#else
    __shared__ __attribute__((aligned(16))) float A[DP * LD];
#endif
    __shared__ __attribute__((aligned(16))) float Tblk[16];          // the panel's updated 4 x 4 diagonal block
    __shared__ __attribute__((aligned(16))) float sdiag[VP];
    __shared__ __attribute__((aligned(16))) float dvec[VP];
    __shared__ __attribute__((aligned(16))) float avec[VP];
    __shared__ __attribute__((aligned(16))) float diag[VP];          // L_ii and 1 / L_ii (the diagonal of A keeps C_ii)
    __shared__ __attribute__((aligned(16))) float rdiag[VP];
    __shared__ float inputs[MVN_NIN_PAD];
    __shared__ float red[4];
    const int tid = threadIdx.x, h = tid & 1, pr = tid >> 1;
    const uint32_t n = blockIdx.x;
    if (n >= G.n_local) return;

    // ---- inputs of the covariance expression, d = x - m, zeroed matrices
    if (tid < MVN_NIN) {
        float v;
        if (tid < MVN_NSI) {
            v = G.samples[(size_t)G.input_rows[tid] * G.n_local + n];
        } else {
            const bsvi_uniform_entry e = G.uniform_inputs[tid - MVN_NSI];
            v = e.a + e.b * utransform(e.transform, G.params[e.src]);
        }
        inputs[tid] = v;
    }
    for (int i = tid; i < VP; i += MVN_THREADS) {
        float v = 0.0f;
        if (i < D) {
            float x = MVN_VALUE_LATENT ? G.samples[(size_t)(G.value_row0 + i) * G.n_local + n] : G.vecs[D + i];
            if (MVN_VALUE_PARAM) {
                const bsvi_uniform_entry e = G.value_entries[i];
                x = e.a + e.b * utransform(e.transform, G.params[e.src]);
            }
            float m = G.vecs[i];
            if (MVN_LOC_PARAM) {
                const bsvi_uniform_entry e = G.loc_entries[i];
                m = e.a + e.b * utransform(e.transform, G.params[e.src]);
            }
            v = x - m;
        }
        dvec[i] = v;
        avec[i] = 0.0f;
        diag[i] = 1.0f;
        rdiag[i] = 1.0f;
    }
    for (int i = tid; i < DP * LD / 4; i += MVN_THREADS) {
        reinterpret_cast<mvn_f4*>(A)[i] = mvn_f4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    __syncthreads();
    float in[MVN_NIN_PAD];
#pragma unroll
    for (int k = 0; k < MVN_NIN_PAD; ++k) in[k] = k < MVN_NIN ? inputs[k] : 0.0f;

    // ---- 1: the lower triangle of C
    constexpr int NTRI = D * (D + 1) / 2;
    for (int e = tid; e < NTRI; e += MVN_THREADS) {
        int i, j;
        mvn_tri(e, i, j);
        float c, dc[MVN_NIN_PAD];
        mvn_cov(i, j, in, G.mats, c, dc);
        A[i * LD + j] = c;
    }
    if (tid < DP - D) A[(D + tid) * LD + D + tid] = 1.0f;
    __syncthreads();
    MVN_STEP_END(1)

#if MVN_FORM == 2
    // ---- precision form: alpha = P d from the lower triangle (row part along k, column part down the rows), before P is factorised
    float quad = 0.0f;
    for (int i = tid; i < D; i += MVN_THREADS) {
        float acc = 0.0f;
        for (int k = 0; k <= i; ++k) acc += A[i * LD + k] * dvec[k];
        for (int k = i + 1; k < D; ++k) acc += A[k * LD + i] * dvec[k];
        avec[i] = acc;
        quad += acc * dvec[i];
    }
    quad = mvn_block_sum(quad, red, tid);
#endif

    // ---- 2: Cholesky (left-looking): column j from the columns before it
    float logdet = 0.0f;
#if MVN_FORM == 1
    // (scale_tril: the lower triangle already holds L)
    {
        float part = 0.0f;
        for (int i = tid; i < D; i += MVN_THREADS) {
            const float l = A[i * LD + i];
            diag[i] = l;
            rdiag[i] = 1.0f / l;
            part += logf(l);
        }
        logdet = mvn_block_sum(part, red, tid);
    }
#else
    // Panels of FOUR columns (one 16-byte block): per panel ONE pass over the row's blocks in front of it — the row's block against
    // the same blocks of the panel's four rows, 16 multiply-adds per five loads where the column-by-column form had 8 per two —
    // then the 4 x 4 diagonal block goes through LDS, every thread factorises it in registers and solves its own row's four
    // entries.  Two barriers per panel = D / 2 in all, and D / 4 dependent passes, where one column at a time took D of each
    // (tools/r4/mvn_steps.sh: 1 875 cycles per column, nearly all of it the latency of that one chain).
    constexpr int TRIPS = (DP + MVN_THREADS / 2 - 1) / (MVN_THREADS / 2);      // rows per pair: 1, or 2 when DP > 128
    for (int c0 = 0; c0 < DP; c0 += 4) {
        const int nb = c0 >> 2;
        const mvn_f4* const y0 = reinterpret_cast<const mvn_f4*>(A + (c0 + 0) * LD);
        const mvn_f4* const y1 = reinterpret_cast<const mvn_f4*>(A + (c0 + 1) * LD);
        const mvn_f4* const y2 = reinterpret_cast<const mvn_f4*>(A + (c0 + 2) * LD);
        const mvn_f4* const y3 = reinterpret_cast<const mvn_f4*>(A + (c0 + 3) * LD);
        float t[TRIPS][4];
#pragma unroll
        for (int tt = 0; tt < TRIPS; ++tt) {
            const int i = c0 + pr + tt * (MVN_THREADS / 2);
            t[tt][0] = t[tt][1] = t[tt][2] = t[tt][3] = 0.0f;
            if (i < DP) {                                     // (a pair's lanes share i)
                const mvn_f4* const ri = reinterpret_cast<const mvn_f4*>(A + i * LD);
                const mvn_f4 own = ri[nb];                   // C_i,c0..c0+3 (zeros above the diagonal)
                float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f, u0 = 0.0f, u1 = 0.0f, u2 = 0.0f, u3 = 0.0f;
                int b = h;
                for (; b + 2 < nb; b += 4) {
                    const mvn_f4 xa = ri[b], xb = ri[b + 2];
                    const mvn_f4 a0 = y0[b], a1 = y1[b], a2 = y2[b], a3 = y3[b];
                    const mvn_f4 b0 = y0[b + 2], b1 = y1[b + 2], b2 = y2[b + 2], b3 = y3[b + 2];
                    t0 += mvn_dot4(xa, a0); t1 += mvn_dot4(xa, a1); t2 += mvn_dot4(xa, a2); t3 += mvn_dot4(xa, a3);
                    u0 += mvn_dot4(xb, b0); u1 += mvn_dot4(xb, b1); u2 += mvn_dot4(xb, b2); u3 += mvn_dot4(xb, b3);
                }
                if (b < nb) {
                    const mvn_f4 xa = ri[b];
                    t0 += mvn_dot4(xa, y0[b]); t1 += mvn_dot4(xa, y1[b]); t2 += mvn_dot4(xa, y2[b]); t3 += mvn_dot4(xa, y3[b]);
                }
                t0 += u0; t1 += u1; t2 += u2; t3 += u3;
                t0 += mvn_pair(t0); t1 += mvn_pair(t1); t2 += mvn_pair(t2); t3 += mvn_pair(t3);
                t[tt][0] = own.x - t0; t[tt][1] = own.y - t1; t[tt][2] = own.z - t2; t[tt][3] = own.w - t3;
                if (h == 0 && i - c0 < 4) *reinterpret_cast<mvn_f4*>(Tblk + 4 * (i - c0)) = mvn_f4{t[tt][0], t[tt][1], t[tt][2], t[tt][3]};
            }
        }
        __syncthreads();
        // the diagonal block T = L_pp L_pp^T, by every thread alike
        const mvn_f4 T0 = reinterpret_cast<const mvn_f4*>(Tblk)[0], T1 = reinterpret_cast<const mvn_f4*>(Tblk)[1],
                     T2 = reinterpret_cast<const mvn_f4*>(Tblk)[2], T3 = reinterpret_cast<const mvn_f4*>(Tblk)[3];
        const float r0 = mvn_rsqrt(T0.x), l00 = T0.x * r0;           // (a pivot <= 0 gives NaN: the step is then skipped as non-finite)
        const float l10 = T1.x * r0;
        const float p1 = T1.y - l10 * l10, r1 = mvn_rsqrt(p1), l11 = p1 * r1;
        const float l20 = T2.x * r0, l21 = (T2.y - l20 * l10) * r1;
        const float p2 = T2.z - l20 * l20 - l21 * l21, r2 = mvn_rsqrt(p2), l22 = p2 * r2;
        const float l30 = T3.x * r0, l31 = (T3.y - l30 * l10) * r1, l32 = (T3.z - l30 * l20 - l31 * l21) * r2;
        const float p3 = T3.w - l30 * l30 - l31 * l31 - l32 * l32, r3 = mvn_rsqrt(p3), l33 = p3 * r3;
#pragma unroll
        for (int tt = 0; tt < TRIPS; ++tt) {
            const int i = c0 + pr + tt * (MVN_THREADS / 2), r = i - c0;
            // this row's four entries: L_i,c0.. = t L_pp^-T (the rows of the block itself: the block's own row, zeros above the diagonal)
            float x0 = t[tt][0] * r0;
            float x1 = (t[tt][1] - x0 * l10) * r1;
            float x2 = (t[tt][2] - x0 * l20 - x1 * l21) * r2;
            float x3 = (t[tt][3] - x0 * l30 - x1 * l31 - x2 * l32) * r3;
            if (r == 0) { x0 = l00; x1 = 0.0f; x2 = 0.0f; x3 = 0.0f; }
            if (r == 1) { x0 = l10; x1 = l11; x2 = 0.0f; x3 = 0.0f; }
            if (r == 2) { x0 = l20; x1 = l21; x2 = l22; x3 = 0.0f; }
            if (r == 3) { x0 = l30; x1 = l31; x2 = l32; x3 = l33; }
            if (h == 0 && i < DP) *reinterpret_cast<mvn_f4*>(A + i * LD + c0) = mvn_f4{x0, x1, x2, x3};
        }
        if (tid == 0) {
            *reinterpret_cast<mvn_f4*>(diag + c0) = mvn_f4{l00, l11, l22, l33};
            *reinterpret_cast<mvn_f4*>(rdiag + c0) = mvn_f4{r0, r1, r2, r3};
        }
        __syncthreads();
    }
    {   // log det L from the diagonal, once
        float part = 0.0f;
        for (int i = tid; i < D; i += MVN_THREADS) part += logf(diag[i]);
        logdet = mvn_block_sum(part, red, tid);
    }
#endif
    MVN_STEP_END(2)

    // ---- 3: X = L^-1, column j by lane pair j, kept transposed: XT[j][i] = X[i][j] = -(sum_{j<=k<i} L[i][k] X[k][j]) / L[i][i]
    // Row blocks of FOUR: the sums over the blocks in front of a row block are four products sharing the column's loads, the
    // block's own triangle runs in registers, the four results leave as one 16-byte store — D / 4 dependent steps per column.
    // Column j of X is ROW j of A behind its diagonal; the row's entries in front of it are still L.
    for (int j = pr; j < D; j += MVN_THREADS / 2) {
        float* const xj = A + j * LD;
        const mvn_f4* const xr = reinterpret_cast<const mvn_f4*>(xj);
        const int jb = j >> 2, q0 = j & 3;
        for (int I = jb; I < DP / 4; ++I) {
            const int i0 = 4 * I;
            const mvn_f4* const l0 = reinterpret_cast<const mvn_f4*>(A + (i0 + 0) * LD);
            const mvn_f4* const l1 = reinterpret_cast<const mvn_f4*>(A + (i0 + 1) * LD);
            const mvn_f4* const l2 = reinterpret_cast<const mvn_f4*>(A + (i0 + 2) * LD);
            const mvn_f4* const l3 = reinterpret_cast<const mvn_f4*>(A + (i0 + 3) * LD);
            // the block's own triangle of L and the reciprocals of its diagonal: requested in front of the pass
            const mvn_f4 d1 = l1[I], d2 = l2[I], d3 = l3[I];
            const mvn_f4 rd = *reinterpret_cast<const mvn_f4*>(rdiag + i0);
            // s_r = sum_{j <= k < 4 I} L[i0 + r][k] X[k][j]: the column's first block (lane h = 0) without what is L in it
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, u0 = 0.0f, u1 = 0.0f, u2 = 0.0f, u3 = 0.0f;
            if (h == 0 && jb < I) {
                mvn_f4 xa = xr[jb];
                xa.x = q0 > 0 ? 0.0f : xa.x; xa.y = q0 > 1 ? 0.0f : xa.y; xa.z = q0 > 2 ? 0.0f : xa.z;
                s0 = mvn_dot4(xa, l0[jb]); s1 = mvn_dot4(xa, l1[jb]); s2 = mvn_dot4(xa, l2[jb]); s3 = mvn_dot4(xa, l3[jb]);
            }
            int b = jb + 1 + h;
            for (; b + 2 < I; b += 4) {
                const mvn_f4 xa = xr[b], xb = xr[b + 2];
                const mvn_f4 a0 = l0[b], a1 = l1[b], a2 = l2[b], a3 = l3[b];
                const mvn_f4 b0 = l0[b + 2], b1 = l1[b + 2], b2 = l2[b + 2], b3 = l3[b + 2];
                s0 += mvn_dot4(xa, a0); s1 += mvn_dot4(xa, a1); s2 += mvn_dot4(xa, a2); s3 += mvn_dot4(xa, a3);
                u0 += mvn_dot4(xb, b0); u1 += mvn_dot4(xb, b1); u2 += mvn_dot4(xb, b2); u3 += mvn_dot4(xb, b3);
            }
            if (b < I) {
                const mvn_f4 xa = xr[b];
                s0 += mvn_dot4(xa, l0[b]); s1 += mvn_dot4(xa, l1[b]); s2 += mvn_dot4(xa, l2[b]); s3 += mvn_dot4(xa, l3[b]);
            }
            s0 += u0; s1 += u1; s2 += u2; s3 += u3;
            s0 += mvn_pair(s0); s1 += mvn_pair(s1); s2 += mvn_pair(s2); s3 += mvn_pair(s3);
            // rows i0 .. i0 + 3 in order: X[i][j] = -(s + the block's own part) / L_ii; row j itself is 1 / L_jj
            const int q = j - i0;                             // (>= 0 only in the column's first block)
            const float x0 = q > 0 ? 0.0f : (q == 0 ? rd.x : -s0 * rd.x);
            const float x1 = q > 1 ? 0.0f : (q == 1 ? rd.y : -(s1 + d1.x * x0) * rd.y);
            const float x2 = q > 2 ? 0.0f : (q == 2 ? rd.z : -(s2 + d2.x * x0 + d2.y * x1) * rd.z);
            const float x3 = q == 3 ? rd.w : -(s3 + d3.x * x0 + d3.y * x1 + d3.z * x2) * rd.w;
            if (h == 0) {
                if (I > jb) {
                    *reinterpret_cast<mvn_f4*>(xj + i0) = mvn_f4{x0, x1, x2, x3};
                } else {                                       // the first block: what lies in front of the diagonal stays L
                    if (q <= 0) xj[i0] = x0;
                    if (q <= 1) xj[i0 + 1] = x1;
                    if (q <= 2) xj[i0 + 2] = x2;
                    xj[i0 + 3] = x3;
                }
            }
        }
    }
    __syncthreads();
    MVN_STEP_END(3)

#if MVN_FORM == 1
    // ---- scale_tril form: y = L^-1 d = X d, y_j = sum_{k <= j} X[j][k] d_k = sum_k XT[k][j] d_k (lanes along j: conflict-free)
    __shared__ __attribute__((aligned(16))) float yvec[VP];
    for (int j = tid; j < D; j += MVN_THREADS) {
        float acc = 0.0f;
        for (int k = 0; k <= j; ++k) acc += A[k * LD + j] * dvec[k];               // (X[j][k] = A[k][j] above the diagonal, X_jj on it)
        yvec[j] = acc;
    }
    __syncthreads();
#endif

    // ---- 4: S = X^T X:  S[i][j] = sum_{k >= i} X[k][i] X[k][j]  (i >= j): rows i and j of A from block i / 4 on, row i without what is
    //         L in its first block.  S goes where L was (nobody reads L any more); its diagonal, which X_ii still occupies, to sdiag.
    for (int e = tid; e < NTRI; e += MVN_THREADS) {
        int i, j;
        mvn_tri(e, i, j);
        const mvn_f4* const xi = reinterpret_cast<const mvn_f4*>(A + i * LD);
        const mvn_f4* const xq = reinterpret_cast<const mvn_f4*>(A + j * LD);
        const int ib = i >> 2, q0 = i & 3;
        mvn_f4 first = xi[ib];
        first.x = q0 > 0 ? 0.0f : first.x; first.y = q0 > 1 ? 0.0f : first.y; first.z = q0 > 2 ? 0.0f : first.z;
        const float acc = mvn_dot4(first, xq[ib]) + mvn_rowdot(xi, xq, ib + 1, DP / 4, 1);
        if (i == j) sdiag[i] = acc; else A[i * LD + j] = acc;
    }
    __syncthreads();
    MVN_STEP_END(4)

    // ---- 5: alpha = S d, the quadratic form, log p
#if MVN_FORM != 2
    float quad = 0.0f;
    for (int i = tid; i < D; i += MVN_THREADS) {
        // S is symmetric and only its lower triangle is stored: the row in front of the diagonal, the column behind it
        float a0 = sdiag[i] * dvec[i], a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int k = 0;
        for (; k + 3 < i; k += 4) {
            const float* const row = A + i * LD + k;
            a0 += row[0] * dvec[k]; a1 += row[1] * dvec[k + 1]; a2 += row[2] * dvec[k + 2]; a3 += row[3] * dvec[k + 3];
        }
        for (; k < i; ++k) a0 += A[i * LD + k] * dvec[k];
        k = i + 1;
        for (; k + 3 < D; k += 4) {
            a0 += A[k * LD + i] * dvec[k]; a1 += A[(k + 1) * LD + i] * dvec[k + 1];
            a2 += A[(k + 2) * LD + i] * dvec[k + 2]; a3 += A[(k + 3) * LD + i] * dvec[k + 3];
        }
        for (; k < D; ++k) a0 += A[k * LD + i] * dvec[k];
        const float acc = (a0 + a1) + (a2 + a3);
        avec[i] = acc;
        quad += acc * dvec[i];
    }
    quad = mvn_block_sum(quad, red, tid);
    const float logp = -0.5f * quad - logdet - 0.5f * (float)D * 1.8378770664093453f;
#else
    const float logp = -0.5f * quad + logdet - 0.5f * (float)D * 1.8378770664093453f;      // (logdet = sum log M_ii = log det P / 2)
#endif
    MVN_STEP_END(5)

    // ---- 6: dlogp/ds_k = sum_{i >= j} m_ij (alpha_i alpha_j - S_ij) / 2 * dC_ij/ds_k   (m_ij = 2 off the diagonal: C is symmetric)
    float gin[MVN_NIN_PAD];
#pragma unroll
    for (int k = 0; k < MVN_NIN_PAD; ++k) gin[k] = 0.0f;
#if MVN_NIN > 0
    for (int e = tid; e < NTRI; e += MVN_THREADS) {
        int i, j;
        mvn_tri(e, i, j);
        float c, dc[MVN_NIN_PAD];
        mvn_cov(i, j, in, G.mats, c, dc);
#if MVN_FORM == 0
        const float sij = i == j ? sdiag[i] : A[i * LD + j];
        const float gij = (i == j ? 0.5f : 1.0f) * (avec[i] * avec[j] - sij);
#elif MVN_FORM == 1
        const float gij = avec[i] * yvec[j] - (i == j ? rdiag[i] : 0.0f);                 // (X_ii = 1 / L_ii)
#else
        const float sij = i == j ? sdiag[i] : A[i * LD + j];
        const float gij = (i == j ? 0.5f : 1.0f) * (sij - dvec[i] * dvec[j]);
#endif
#pragma unroll
        for (int k = 0; k < MVN_NIN; ++k) gin[k] += gij * dc[k];
    }
#pragma unroll
    for (int k = 0; k < MVN_NIN; ++k) gin[k] = mvn_block_sum(gin[k], red, tid);
#endif

    // ---- the surrogate's rows: coefficients of the slot inputs, of x (when latent), of the uniform inputs, then e
    const float w = G.weight;
    float lin = 0.0f;              // sum_k g_k * input_k over everything that carries a coefficient
#pragma unroll
    for (int k = 0; k < MVN_NIN; ++k) lin += w * gin[k] * in[k];
    float linx = 0.0f;
    uint32_t row = 0;
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < MVN_NSI; ++k) G.rows_out[(size_t)(row + k) * G.n_local + n] = w * gin[k];
    }
    row += MVN_NSI;
    if (MVN_VALUE_LATENT || MVN_VALUE_PARAM) {
        for (int i = tid; i < D; i += MVN_THREADS) {
            const float gx = -w * avec[i];
            G.rows_out[(size_t)(row + i) * G.n_local + n] = gx;
            float x;
            if (MVN_VALUE_PARAM) {
                const bsvi_uniform_entry e = G.value_entries[i];
                x = e.a + e.b * utransform(e.transform, G.params[e.src]);
            } else {
                x = G.samples[(size_t)(G.value_row0 + i) * G.n_local + n];
            }
            linx += gx * x;
        }
        linx = mvn_block_sum(linx, red, tid);
        row += D;
    }
    if (tid == 0) {
#pragma unroll
        for (int k = MVN_NSI; k < MVN_NIN; ++k) G.rows_out[(size_t)(row + k - MVN_NSI) * G.n_local + n] = w * gin[k];
    }
    row += MVN_NIN - MVN_NSI;
    float linm = 0.0f;
    if (MVN_LOC_PARAM) {           // d log p / d m = +alpha
        for (int i = tid; i < D; i += MVN_THREADS) {
            const float gm = w * avec[i];
            G.rows_out[(size_t)(row + i) * G.n_local + n] = gm;
            const bsvi_uniform_entry e = G.loc_entries[i];
            linm += gm * (e.a + e.b * utransform(e.transform, G.params[e.src]));
        }
        linm = mvn_block_sum(linm, red, tid);
        row += D;
    }
    if (tid == 0) G.rows_out[(size_t)row * G.n_local + n] = w * logp - lin - linx - linm;
}

}  // namespace bsvi
