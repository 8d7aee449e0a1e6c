// amort_kernel.hip — the amortised (encoder / decoder network) ELBO path for MI355X (gfx950).
//
// What it replaces in the reference (BASELINE config 5, examples/VAE_playground.py:18-88): per iteration
// `perform_inference` (inference.py:95-108) draws, for EVERY Monte-Carlo sample, its own minibatch of B dataset
// rows (EmpiricalDistribution._get_sample, distributions.py:410-457), runs the encoder module on the N*B rows
// (BrancherFunction -> torch.nn.Module, functions.py:28-41), draws z = loc + scale*eps (distributions.py:111-124),
// runs the decoder module, evaluates Binomial(1, logits) + Normal prior + Normal entropy per row
// (variables.py:851-855) and back-propagates through both modules with autograd.
//
// Here the iteration is a fixed sequence of launches over R = N*B rows:
//   amort_rows        minibatch indices of every sample (keyed bijection, distinct rows per sample)
//   gemm<NT>          one per Linear layer of the encoder: y = act(x W^T + b)   (row gather fused into layer 1)
//   amort_latent_fwd  eps (Philox or supplied), z, log p(z), H[q], log q per row
//   gemm<NT>          decoder layers
//   amort_lik         log p(x | z) per row and dlogits in place (one wave per row)
//   per layer, last to first:  gemm<TN> partial dW = dY^T x per slice of K = R (one slice per workgroup, plain stores
//                              into the workspace; partial db = 1^T dY in the same launch),
//                              gemm<NN> dX = (dY W) * act'(x)
//   reduce_partials   ONE launch at the end: every gradient element = the sum of its slices in slice order — the
//                     gradients (and the loss sums) are bit-reproducible call to call, no float atomics anywhere
//   layers with a side of width <= 8 (latent heads, first decoder layer) use memory-bound kernels instead: rowdot_kernel
//   (16 lanes per row, DPP row sums), outer_kernel (four-wide outer products, eight rows in flight), skinny_k4[nt]_kernel
//   (four outputs per thread); the one-output-per-thread skinny_* kernels are the unaligned fallbacks
//   amort_latent_bwd  joins decoder dz with prior / entropy / score-function terms, loss sums
// GEMMs: 128x128x16 or 64x128x16 workgroup tiles, four waves, v_mfma_f32_32x32x2_f32, k-major LDS tiles (row stride 132
// words: transposing stores and MFMA operand reads are both conflict-free), two register sets of global loads in flight
// three steps ahead of their use, workgroup order remapped so that the tiles sharing rows of the tall operand sit on one XCD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "bsvi.h"
#include "bsvi_internal.h"
#include "philox.h"

namespace bsvi_amort_impl {
using bsvi::philox4x32;
using bsvi::u01;
using bsvi::u32x4;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16, NTHREADS = 256;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

enum { MODE_NT = 0, MODE_NN = 1, MODE_TN = 2 };

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const int32_t* rows;   // gather of the R-indexed rows: A's rows (NT, NN) or B's rows (TN)
    int M, N, K;
    int lda, ldb, ldc;
    int vecA, vecB;        // 16-byte loads allowed (pointer and leading dimension aligned)
    const float* bias;     // NT
    const float* Y;        // NN: value whose producing activation is differentiated
    int ldy;
    int act;
    float post_add;
    int split;             // output columns >= split use act2 / post_add2 (merged sibling layers); 0 = no split
    int act2;
    float post_add2;
    int accumulate;        // NN: C += result
    int k_chunk;           // TN: rows of K per workgroup (multiple of BK)
    int remap;             // XCD-aware workgroup order
    int tiles;             // TN: output tiles per k split
    float* bias_grad;      // TN: partial db[split][m] = sum_k A[k][m], taken from the A tiles the n = 0 workgroups stream anyway
    long part_stride;      // TN: floats between the partial outputs of consecutive k splits (C is the split-0 slice)
};

__device__ __forceinline__ float act_forward(int act, float v, float post_add) {
    if (act == BSVI_ACT_RELU) v = fmaxf(v, 0.0f);
    if (act == BSVI_ACT_SOFTPLUS) v = v > 20.0f ? v : log1pf(expf(v));   // torch softplus, threshold 20
    return v + post_add;
}
// merged sibling layers: the activation of output column n
#define ACT_OF(G, n) (((G).split > 0 && (n) >= (G).split) ? (G).act2 : (G).act)
#define ADD_OF(G, n) (((G).split > 0 && (n) >= (G).split) ? (G).post_add2 : (G).post_add)
// derivative of the activation expressed through its OUTPUT y (what the forward pass kept)
__device__ __forceinline__ float act_derivative(int act, float y, float post_add) {
    if (act == BSVI_ACT_RELU) return y - post_add > 0.0f ? 1.0f : 0.0f;
    if (act == BSVI_ACT_SOFTPLUS) {
        const float sp = y - post_add;                    // softplus(v); sigmoid(v) = 1 - exp(-softplus(v))
        return sp > 20.0f ? 1.0f : -expm1f(-sp);
    }
    return 1.0f;
}

// ---- tile loaders -------------------------------------------------------------------------------------------
// KC: the operand is stored [rows][k] (k contiguous); the tile is rows r0.. x k kt..kt+15, transposed on the way
//     into LDS.  Thread t owns rows (t>>2) and (t>>2)+64, k quad (t&3)*4.
// MC: the operand is stored [k][cols] (cols contiguous); thread t owns k rows (t>>5) and (t>>5)+8, columns (t&31)*4.
template <int W> struct Frag { float4 v[W / 64]; };   // W = tile extent along m / n: 128 or 64

__device__ __forceinline__ float4 load4(const float* p, int valid, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid >= 4 && vec) {
        v = *reinterpret_cast<const float4*>(p);
    } else if (valid > 0) {
        v.x = p[0];
        if (valid > 1) v.y = p[1];
        if (valid > 2) v.z = p[2];
        if (valid > 3) v.w = p[3];
    }
    return v;
}

// KC: rows (t>>2) + 64 i, k quad (t&3)*4
template <int W>
__device__ __forceinline__ Frag<W> load_kc(const float* base, const int32_t* rows, int ld, int r0, int n_rows, int kt,
                                           int k_end, bool vec) {
    Frag<W> f;
    const int t = threadIdx.x, kq = kt + (t & 3) * 4;
#pragma unroll
    for (int i = 0; i < W / 64; ++i) {
        const int r = r0 + (t >> 2) + 64 * i;
        int valid = 0;
        const float* p = base;
        if (r < n_rows) {
            const long phys = rows ? (long)rows[r] : (long)r;
            p = base + phys * ld + kq;
            valid = k_end - kq;
        }
        f.v[i] = load4(p, valid, vec);
    }
    return f;
}
template <int W>
__device__ __forceinline__ void store_kc(float* tile, const Frag<W>& f) {
    constexpr int LD = W + 4;
    const int t = threadIdx.x, kq = (t & 3) * 4;
#pragma unroll
    for (int i = 0; i < W / 64; ++i) {
        const int r = (t >> 2) + 64 * i;
        tile[(kq + 0) * LD + r] = f.v[i].x;
        tile[(kq + 1) * LD + r] = f.v[i].y;
        tile[(kq + 2) * LD + r] = f.v[i].z;
        tile[(kq + 3) * LD + r] = f.v[i].w;
    }
}
// MC: W/4 threads per k row, 1024/W k rows per pass, W/64 passes
template <int W>
__device__ __forceinline__ Frag<W> load_mc(const float* base, const int32_t* rows, int ld, int c0, int n_cols, int kt,
                                           int k_end, bool vec) {
    constexpr int TPR = W / 4, KR = NTHREADS / TPR;
    Frag<W> f;
    const int t = threadIdx.x, c = c0 + (t % TPR) * 4;
#pragma unroll
    for (int i = 0; i < W / 64; ++i) {
        const int k = kt + t / TPR + KR * i;
        int valid = 0;
        const float* p = base;
        if (k < k_end) {
            const long phys = rows ? (long)rows[k] : (long)k;
            p = base + phys * ld + c;
            valid = n_cols - c;
        }
        f.v[i] = load4(p, valid, vec);
    }
    return f;
}
template <int W>
__device__ __forceinline__ void store_mc(float* tile, const Frag<W>& f) {
    constexpr int TPR = W / 4, KR = NTHREADS / TPR, LD = W + 4;
    const int t = threadIdx.x, c = (t % TPR) * 4;
#pragma unroll
    for (int i = 0; i < W / 64; ++i) {
        const int k = t / TPR + KR * i;
        // (built from the components: copying the float4 object itself through the cast makes the compiler keep the whole
        //  fragment in scratch memory — a scratch store + load per step in the weight- and input-gradient kernels)
        *reinterpret_cast<f32x4*>(&tile[k * LD + c]) = f32x4{f.v[i].x, f.v[i].y, f.v[i].z, f.v[i].w};
    }
}

// TBM x 128 workgroup tile (TBM = 128: waves 2x2 of 64x64; TBM = 64: waves 2x2 of 32x64 — twice the workgroups when
// the tall operand alone does not fill the chip)
template <int MODE, int TBM, int PF>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmArgs G) {
    constexpr int LDA = TBM + 4, LDB = BN + 4, TM = TBM / 64;
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

    const int tiles_n = (G.N + BN - 1) / BN;
    const int n_blocks = gridDim.x;
    int bid = blockIdx.x;
    if (G.remap) bid = (bid & 7) * (n_blocks >> 3) + (bid >> 3);      // XCD x gets one contiguous range of tiles
    int k_begin = 0, k_end = G.K, split = 0;
    if (MODE == MODE_TN) {
        // the remapped index runs over (k split, tile) with the tile fastest: an XCD owns whole k splits, so the rows
        // of both operands in a split are fetched into ONE L2 (launch order would spread a split's tiles over all
        // eight and every XCD would pull every row through the fabric)
        split = bid / G.tiles;
        bid -= split * G.tiles;
        k_begin = split * G.k_chunk;
        k_end = min(G.K, k_begin + G.k_chunk);
        if (k_begin >= G.K) return;                                   // padding split (grid rounded up to 8)
    }
    const int m0 = (bid / tiles_n) * TBM, n0 = (bid % tiles_n) * BN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * (TBM / 2), wn = (wave & 1) * 64;
    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto load_a = [&](int kt) {
        if (MODE == MODE_TN) return load_mc<TBM>(G.A, nullptr, G.lda, m0, G.M, kt, k_end, G.vecA);
        return load_kc<TBM>(G.A, G.rows, G.lda, m0, G.M, kt, k_end, G.vecA);
    };
    auto load_b = [&](int kt) {
        if (MODE == MODE_NT) return load_kc<BN>(G.B, nullptr, G.ldb, n0, G.N, kt, k_end, G.vecB);
        return load_mc<BN>(G.B, MODE == MODE_TN ? G.rows : nullptr, G.ldb, n0, G.N, kt, k_end, G.vecB);
    };
    auto store_a = [&](float* tile, const Frag<TBM>& f) {
        if (MODE == MODE_TN) store_mc<TBM>(tile, f); else store_kc<TBM>(tile, f);
    };
    auto store_b = [&](float* tile, const Frag<BN>& f) {
        if (MODE == MODE_NT) store_kc<BN>(tile, f); else store_mc<BN>(tile, f);
    };

    // bias gradient on the side (TN, TBM = 128): every thread sums the A values it stages — 4 columns, its k rows
    const bool col_sums = MODE == MODE_TN && G.bias_grad != nullptr && n0 == 0;
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add_cols = [&](const Frag<TBM>& f) {
#pragma unroll
        for (int i = 0; i < TBM / 64; ++i) { bs.x += f.v[i].x; bs.y += f.v[i].y; bs.z += f.v[i].z; bs.w += f.v[i].w; }
    };

    const int n_steps = (k_end - k_begin + BK - 1) / BK;
    const int lk = lane >> 5, lm = lane & 31;

    // Workgroup-uniform fast path: every row is 16-byte loadable and the k range is whole steps -> per-thread
    // pointers are resolved once (row gather included) and a step's staging is 3-4 straight-line global_load_dwordx4.
    // A tile that sticks out of the matrix along m / n stays on this path: the out-of-range rows / column quads are
    // clamped onto valid ones (their products land in outputs the epilogue never stores), so edge tiles are not the
    // stragglers of the launch.  The generic loaders (scalar tails, k tails) take what is left.
    const bool fast = G.vecA && G.vecB && (k_end - k_begin) % BK == 0 && k_end > k_begin;
    const bool gather_b = MODE == MODE_TN && G.rows != nullptr;   // weight gradient of the layer that reads the data rows
    int ridx[BN / 64];
    const float* pa[TBM / 64];
    const float* pb[BN / 64];
    long sa = BK, sb = BK;       // pointer advance per step
    if (fast) {
        const int t = threadIdx.x;
        if (MODE == MODE_TN) {
            constexpr int TPR = TBM / 4, KR = NTHREADS / TPR;
#pragma unroll
            for (int i = 0; i < TBM / 64; ++i)
                pa[i] = G.A + (long)(k_begin + t / TPR + KR * i) * G.lda + min(m0 + (t % TPR) * 4, G.lda - 4);
            sa = (long)BK * G.lda;
        } else {
#pragma unroll
            for (int i = 0; i < TBM / 64; ++i) {
                const int r = min(m0 + (t >> 2) + 64 * i, G.M - 1);
                pa[i] = G.A + (G.rows ? (long)G.rows[r] : (long)r) * G.lda + k_begin + (t & 3) * 4;
            }
        }
        if (MODE == MODE_NT) {
#pragma unroll
            for (int i = 0; i < BN / 64; ++i)
                pb[i] = G.B + (long)min(n0 + (t >> 2) + 64 * i, G.N - 1) * G.ldb + k_begin + (t & 3) * 4;
        } else {
            constexpr int TPR = BN / 4, KR = NTHREADS / TPR;
#pragma unroll
            for (int i = 0; i < BN / 64; ++i) {
                pb[i] = G.B + (long)(k_begin + t / TPR + KR * i) * G.ldb + min(n0 + (t % TPR) * 4, G.ldb - 4);
                if (gather_b) ridx[i] = G.rows[k_begin + t / TPR + KR * i];
            }
            sb = (long)BK * G.ldb;
        }
    }

    auto main_loop = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        auto stage = [&](int step, Frag<TBM>& fa, Frag<BN>& fb) {
            if (FAST) {
#pragma unroll
                for (int i = 0; i < TBM / 64; ++i) fa.v[i] = *reinterpret_cast<const float4*>(pa[i] + step * sa);
                if (gather_b) {
                    // rows of this step were fetched one step ahead, so index and data latencies do not chain
                    constexpr int TPR = BN / 4, KR = NTHREADS / TPR;
                    const int t = threadIdx.x;
#pragma unroll
                    for (int i = 0; i < BN / 64; ++i)
                        fb.v[i] = *reinterpret_cast<const float4*>(G.B + (long)ridx[i] * G.ldb + min(n0 + (t % TPR) * 4, G.ldb - 4));
                    if (step + 1 < n_steps) {
#pragma unroll
                        for (int i = 0; i < BN / 64; ++i) ridx[i] = G.rows[k_begin + (step + 1) * BK + t / TPR + KR * i];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < BN / 64; ++i) fb.v[i] = *reinterpret_cast<const float4*>(pb[i] + step * sb);
                }
            } else {
                fa = load_a(k_begin + step * BK);
                fb = load_b(k_begin + step * BK);
            }
        };
        // PF register sets, two LDS buffers: tile s+1 is written to LDS right AFTER the barrier that frees its buffer — a
        // whole step before it is read, so the ds_write -> barrier -> ds_read chain is off the critical path — and the
        // loads of tile s+1+PF are re-issued into the same registers at once; they stay in flight across PF steps' MFMAs
        // and barriers (plain global loads survive __syncthreads(); the compiler counts vmcnt so that a store waits for
        // its own set only).  The loop is unrolled PF times so that the sets are named registers.
        Frag<TBM> fa[PF];
        Frag<BN> fb[PF];
        stage(0, fa[0], fb[0]);
        if (col_sums) add_cols(fa[0]);
        store_a(As[0], fa[0]);
        store_b(Bs[0], fb[0]);
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (1 + p < n_steps) stage(1 + p, fa[p], fb[p]);      // set p holds step 1 + p (+ multiples of PF later)
        __syncthreads();
        auto one_step = [&](int step, Frag<TBM>& ra, Frag<BN>& rb) {
            const int cur = step & 1;
            if (step + 1 < n_steps) {
                if (col_sums) add_cols(ra);
                store_a(As[cur ^ 1], ra);
                store_b(Bs[cur ^ 1], rb);
                if (step + 1 + PF < n_steps) stage(step + 1 + PF, ra, rb);
            }
            // all operand fragments of a batch first (immediate-offset ds_reads off one base per tile), then the
            // MFMAs back to back: the matrix pipe is not stalled on an LDS round trip every second instruction
            const float* at = (cur ? As[1] : As[0]) + lk * LDA + wm + lm;
            const float* bt = (cur ? Bs[1] : Bs[0]) + lk * LDB + wn + lm;
#pragma unroll
            for (int h = 0; h < 2; ++h) {       // two batches of four k pairs: 12-16 live operand registers, not 24-32
                float a[BK / 4][TM], b[BK / 4][2];
#pragma unroll
                for (int kk = 0; kk < BK / 4; ++kk) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[kk][i] = at[(BK / 2 * h + 2 * kk) * LDA + 32 * i];
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[kk][j] = bt[(BK / 2 * h + 2 * kk) * LDB + 32 * j];
                }
#pragma unroll
                for (int kk = 0; kk < BK / 4; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        };
        for (int step = 0; step < n_steps; step += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p)
                if (step + p < n_steps) one_step(step + p, fa[p], fb[p]);   // step s reads set (s mod PF): holds step s + 1
        }
    };
    if (fast) main_loop(std::true_type{}); else main_loop(std::false_type{});

    if (col_sums) {   // fold the k-row groups of the workgroup (MC loader: TBM/4 threads per k row) through LDS
        constexpr int TPR = TBM / 4, KR = NTHREADS / TPR;
        float* red = As[0];
        *reinterpret_cast<float4*>(&red[(threadIdx.x / TPR) * LDA + (threadIdx.x % TPR) * 4]) = bs;
        __syncthreads();
        if (threadIdx.x < TBM && m0 + (int)threadIdx.x < G.M) {
            float s = 0.0f;
#pragma unroll
            for (int q = 0; q < KR; ++q) s += red[q * LDA + threadIdx.x];
            G.bias_grad[(long)split * G.M + m0 + threadIdx.x] = s;     // every (split, m) has exactly one writer
        }
    }

    // epilogue.  acc[i][j][r] is C[m][n] with m = 32i + 8(r>>2) + 4(lane>>5) + (r&3), n = 32j + (lane&31)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + lm;
            if (n >= G.N) continue;
            float bias = 0.0f;
            if (MODE == MODE_NT && G.bias) bias = G.bias[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * lk + (r & 3);
                if (m >= G.M) continue;
                float v = acc[i][j][r];
                float* c = G.C + (long)m * G.ldc + n;
                if (MODE == MODE_NT) {
                    *c = act_forward(ACT_OF(G, n), v + bias, ADD_OF(G, n));
                } else if (MODE == MODE_NN) {
                    if (G.Y) v *= act_derivative(ACT_OF(G, n), G.Y[(long)m * G.ldy + n], ADD_OF(G, n));
                    *c = G.accumulate ? *c + v : v;
                } else {
                    c[(long)split * G.part_stride] = v;                    // partial of this k split, summed by reduce_partials
                }
            }
        }
}

// ---- the layer that reads the DATA rows, on the bf16 matrix cores at f32 accuracy ------------------------------------
// examples/VAE_playground.py:22-26 binarises its images; pixel counts 0..255 and {0, 1} alike are EXACTLY representable
// in bf16 (8 significant bits).  With x exact, a product x*w with w split into three bf16 pieces w = hi + mid + lo (8 + 8 +
// 8 significant bits: the split of an f32 is exact) is the sum of three bf16 products, each exact in the f32 accumulator —
// so x W^T runs on v_mfma_f32_32x32x16_bf16 (16x the rate of the f32-input MFMA) as THREE MFMAs per tile and k step, with
// the rounding behaviour of an f32 fma chain (one rounding per accumulation, summation order hi, mid, lo within a k step).
// bsvi_amort_create checks every dataset value for exactness and keeps a bf16 copy of the dataset, rows padded with zeros
// to a multiple of 32 columns; xw_split_kernel splits the layer's weights once per iteration (they change every step).
// A dataset with any inexact value runs on the f32 kernel above.  BSVI_AMORT_XGEMM=0 forces that too.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t xu4 __attribute__((ext_vector_type(4)));   // (ext vectors, not HIP's uint4 struct: copies of that through pointer casts land in scratch)
constexpr int XBK = 32;                 // k per step: two MFMA k chunks of 16
constexpr int XLD = 80;                 // LDS row stride in bytes (64 data + 16 pad: 16 consecutive rows hit 16 disjoint bank quads)
constexpr int XPLANE = 128 * XLD;       // one 128-row operand tile

struct XGemmArgs {
    const uint16_t* X;      // [DS][Kp] bf16 dataset copy
    const int32_t* rows;    // gather of the M rows (null: identity)
    const uint16_t* Wp;     // [3][N][Kp] bf16 pieces of the weights
    long plane_stride;      // elements between pieces
    float* C;
    const float* bias;
    int M, N, Kp, ldc;
    int act; float post_add; int split; int act2; float post_add2;
    int rows_fastest;       // tile order: neighbours share a column tile (the pieces) instead of a row tile (X)
    // the split-k form (weight gradient of the data layer): workgroup (tile, split) multiplies k steps
    // [split * steps_per_split, ...) and writes its partial TRANSPOSED, C + split * part_stride as [N][ldc] with m along a row
    int steps_per_split;
    long part_stride;
};

__device__ __forceinline__ uint16_t bf16_bits(float x) { const __bf16 b = (__bf16)x; return __builtin_bit_cast(uint16_t, b); }
__device__ __forceinline__ float bf16_value(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }

// W [N][K] f32 -> three bf16 pieces [3][N][Kp] (zero beyond K): hi = rne(w), mid = rne(w - hi), lo = rne(w - hi - mid)
__global__ __launch_bounds__(256) void xw_split_kernel(const float* W, int N, int K, int Kp, uint16_t* Wp) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * Kp) return;
    const int n = (int)(i / Kp), k = (int)(i - (long)n * Kp);
    const float w = k < K ? W[(long)n * K + k] : 0.0f;
    const uint16_t hi = bf16_bits(w);
    const float r1 = w - bf16_value(hi);
    const uint16_t mid = bf16_bits(r1);
    const float r2 = r1 - bf16_value(mid);
    const long plane = (long)N * Kp;
    Wp[i] = hi; Wp[plane + i] = mid; Wp[2 * plane + i] = bf16_bits(r2);
}

// f32 rows -> bf16 rows padded to Kp (tests / the debug hook; the product path converts the dataset once on the host)
__global__ __launch_bounds__(256) void x_to_bf16_kernel(const float* X, long rows, int K, int Kp, uint16_t* Xb) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * Kp) return;
    const long r = i / Kp;
    const int k = (int)(i - r * Kp);
    Xb[i] = k < K ? bf16_bits(X[r * K + k]) : (uint16_t)0;
}

// C[M][N] = act(X[rows[m]] W^T + bias).  Workgroup tile TBM x 128 (TBM = 256: four waves of 128 x 64; 128: of 64 x 64), single
// LDS stage with the next step's global loads in flight behind the MFMAs.  The kernel is LDS-bandwidth bound — per k chunk
// a wave reads FA fragments of X and 3 x 2 of the pieces for 3 x FA x 2 MFMAs — so the exact operand gets the long side of
// the wave tile: 0.42 fragment reads per MFMA at FA = 4 against 0.67 at FA = 2 (cfg 5's first layer 55 -> see DESIGN 4.6).
template <int TBM>
__global__ __launch_bounds__(256, 2) void xgemm_nt_kernel(const XGemmArgs G) {
    constexpr int FA = TBM / 64, APLANE = TBM * XLD, NA = TBM * 4 / 256;      // X fragments per wave; A stage bytes; A pieces per thread
    __shared__ __attribute__((aligned(16))) unsigned char lds[APLANE + 3 * XPLANE];      // X | W hi | W mid | W lo
    const int tiles_n = (G.N + 127) / 128;
    int bid = blockIdx.x;
    const int n_blocks = gridDim.x;
    if ((n_blocks & 7) == 0) bid = (bid & 7) * (n_blocks >> 3) + (bid >> 3);      // an XCD takes a contiguous range of tiles
    // ... in which the tiles that share the LARGER operand are neighbours, so that its rows enter one L2 once: with the
    // pieces of cfg 4's W (49 MB, four row tiles of the minibatch above each column tile) in column-fastest order every XCD
    // pulled half of them through the fabric — 196 MB per product
    const int tiles_m = (G.M + TBM - 1) / TBM;
    int tm, tn;
    if (G.rows_fastest) { tm = bid % tiles_m; tn = bid / tiles_m; } else { tm = bid / tiles_n; tn = bid % tiles_n; }
    const int m0 = tm * TBM, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (TBM / 2), wn = (wave & 1) * 64, lm = lane & 31, lk = lane >> 5;

    // staging: 16-byte pieces.  X: TBM * 4 per step (NA per thread), W: 1536 (6 per thread)
    // (byte offsets from the two base pointers, 32 bits each: the operands are far below 4 GB, and ten 64-bit pointers per
    //  thread were the difference between one and two waves per SIMD at TBM = 256)
    const unsigned char* const xbase = reinterpret_cast<const unsigned char*>(G.X);
    const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(G.Wp);
    uint32_t pa[NA], pb[6];
    int sa[NA], sb[6];       // LDS byte offsets
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = tid + 256 * i, row = c >> 2, kq = c & 3;
        const int r = min(m0 + row, G.M - 1);
        const long phys = G.rows ? (long)G.rows[r] : (long)r;
        pa[i] = (uint32_t)((phys * G.Kp + kq * 8) * 2);
        sa[i] = row * XLD + kq * 16;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = tid + 256 * i, plane = c >> 9, w = c & 511, col = w >> 2, kq = w & 3;
        pb[i] = (uint32_t)((plane * G.plane_stride + (long)min(n0 + col, G.N - 1) * G.Kp + kq * 8) * 2);
        sb[i] = APLANE + plane * XPLANE + col * XLD + kq * 16;
    }
    auto ldx = [&](uint32_t off, int step) { return *reinterpret_cast<const xu4*>(xbase + off + (size_t)step * (XBK * 2)); };
    auto ldw = [&](uint32_t off, int step) { return *reinterpret_cast<const xu4*>(wbase + off + (size_t)step * (XBK * 2)); };
    f32x16 acc[FA][2];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // 32-row blocks of this wave that lie inside the matrix (a tile at the bottom edge: the rest is neither read nor multiplied)
    int live = 0;
#pragma unroll
    for (int i = 0; i < FA; ++i) live += (m0 + wm + 32 * i < G.M) ? 1 : 0;

    const int n_steps = G.Kp / XBK;
    xu4 ra[NA], rb[6];
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = ldx(pa[i], 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) rb[i] = ldw(pb[i], 0);
    const unsigned char* at = lds + (wm + lm) * XLD + lk * 16;
    const unsigned char* bt = lds + APLANE + (wn + lm) * XLD + lk * 16;
    for (int step = 0; step < n_steps; ++step) {
        __syncthreads();                                   // the last step's reads of the stage are done
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<xu4*>(lds + sa[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<xu4*>(lds + sb[i]) = rb[i];
        __syncthreads();
        if (step + 1 < n_steps) {                          // (XBK bf16 = 4 pieces of 16 bytes per row and step)
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = ldx(pa[i], step + 1);
#pragma unroll
            for (int i = 0; i < 6; ++i) rb[i] = ldw(pb[i], step + 1);
        }
        if (live == 0) continue;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 a[FA];
#pragma unroll
            for (int i = 0; i < FA; ++i) a[i] = *reinterpret_cast<const bf16x8*>(at + 32 * i * XLD + kc * 32);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                bf16x8 b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bt + p * XPLANE + 32 * j * XLD + kc * 32);
#pragma unroll
                for (int i = 0; i < FA; ++i) {
                    if (i < live) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    }
    // epilogue: acc[i][j][r] is C[m][n] with m = 32i + 8(r>>2) + 4(lane>>5) + (r&3), n = 32j + (lane&31)   (as gemm_kernel)
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + lm;
            if (n >= G.N) continue;
            const float bias = G.bias ? G.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * lk + (r & 3);
                if (m >= G.M) continue;
                G.C[(long)m * G.ldc + n] = act_forward(ACT_OF(G, n), acc[i][j][r] + bias, ADD_OF(G, n));
            }
        }
}

// The same product with LDS-DMA staging (`global_load_lds_dwordx4`: global -> LDS without registers and without the
// ds_write pass that bounded the kernel above — 13 cycles per ds_write_b128 wave-instruction, ~79 B/clk/CU).  A wave's DMA
// writes 64 x 16 bytes LINEARLY from a wave-uniform base, so an operand tile is stored as 1 KB chunks of 16 rows x 64 bytes
// with no padding, and the bank-conflict-free image is made on the SOURCE side: slot (row r, piece sp) of a chunk holds
// global piece sp ^ ((r >> 2) & 3) of the row; the fragment reads apply the same involution (rows r, r+4, r+8, r+12 of a
// 16-lane ds_read_b128 group then sit on four different bank quads).  Two LDS stages (64 KB per workgroup, two per CU), ONE
// barrier per k step: wait for the DMA of step s, barrier, start the DMA of step s+1 into the other stage, multiply.
// (Round 6, measured and dropped — tools/r6/experiments/xgemm_nt_three_stages.patch, profiles/r6/xgemm_stages_ab.txt: THREE stages, the DMA two
//  steps ahead behind a bare s_barrier: 96 / 120 KB leave one workgroup per CU, and the Bayesian neural network's first product goes from
//  83 to 125 us — the second workgroup of a CU covers more than the deeper prefetch.)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gmem_ptr_t;
template <int TBM, bool SPLITK>
__global__ __launch_bounds__(256, 2) void xgemm_nt_glds_kernel(const XGemmArgs G) {
    constexpr int FA = TBM / 64, NXC = TBM / 64;                          // X fragments per wave; X chunks (of 16 rows) per wave
    constexpr int CHUNK = 1024, APL = 8 * CHUNK, XPL = (TBM / 16) * CHUNK, STAGE = XPL + 3 * APL;      // X (TBM rows) | hi | mid | lo (128 rows each)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
    const int tiles_n = (G.N + 127) / 128, tiles_m = (G.M + TBM - 1) / TBM;
    int bid = blockIdx.x;
    const int n_blocks = gridDim.x;
    if ((n_blocks & 7) == 0) bid = (bid & 7) * (n_blocks >> 3) + (bid >> 3);
    int split = 0;                         // (the tiles of one split are neighbours: they read the same k range of both operands)
    if constexpr (SPLITK) { const int tiles = tiles_m * tiles_n; split = bid / tiles; bid -= split * tiles; }
    int tm, tn;
    if (G.rows_fastest) { tm = bid % tiles_m; tn = bid / tiles_m; } else { tm = bid / tiles_n; tn = bid % tiles_n; }
    const int m0 = tm * TBM, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (TBM / 2), wn = (wave & 1) * 64, lm = lane & 31, lk = lane >> 5;
    const int n_steps = G.Kp / XBK;
    int s_begin = 0, s_end = n_steps;
    if constexpr (SPLITK) {
        s_begin = split * G.steps_per_split;
        s_end = min(n_steps, s_begin + G.steps_per_split);
        if (s_begin >= s_end) return;      // a padding workgroup of the grid
    }

    // DMA sources: wave w fills chunks 2w, 2w+1 of X and chunks 6w .. 6w+5 of the 24 chunks of the pieces.  Lane l of a chunk is
    // slot (row l >> 2, piece l & 3) and loads global piece (l & 3) ^ ((l >> 4) & 3) of that row.
    const unsigned char* const xbase = reinterpret_cast<const unsigned char*>(G.X);
    const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(G.Wp);
    const int gp = (lane & 3) ^ ((lane >> 4) & 3);
    uint32_t src[NXC + 6];
    uint32_t dst[NXC + 6];
#pragma unroll
    for (int i = 0; i < NXC; ++i) {
        const int chunk = NXC * wave + i, row = chunk * 16 + (lane >> 2);
        const int r = min(m0 + row, G.M - 1);
        const long phys = G.rows ? (long)G.rows[r] : (long)r;
        src[i] = (uint32_t)((phys * G.Kp + gp * 8) * 2);
        dst[i] = chunk * CHUNK;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = 6 * wave + i, plane = c >> 3, chunk = c & 7, col = chunk * 16 + (lane >> 2);
        src[NXC + i] = (uint32_t)((plane * G.plane_stride + (long)min(n0 + col, G.N - 1) * G.Kp + gp * 8) * 2);
        dst[NXC + i] = XPL + plane * APL + chunk * CHUNK;
    }
    auto issue = [&](int step, int stage) {
        unsigned char* base = lds + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NXC; ++i)
            __builtin_amdgcn_global_load_lds((gmem_ptr_t)(xbase + src[i] + (size_t)step * (XBK * 2)), (lds_ptr_t)(base + dst[i]), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds((gmem_ptr_t)(wbase + src[NXC + i] + (size_t)step * (XBK * 2)), (lds_ptr_t)(base + dst[NXC + i]), 16, 0, 0);
    };
    f32x16 acc[FA][2];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    int live = 0;                          // 32-row blocks of this wave inside the matrix
#pragma unroll
    for (int i = 0; i < FA; ++i) live += (m0 + wm + 32 * i < G.M) ? 1 : 0;

    // fragment addresses: row R of a plane, piece q -> (R >> 4) * 1 KB + (R & 15) * 64 + ((q ^ ((R >> 2) & 3)) * 16)
    int offa[FA][2], offb[2][2];          // [block][k chunk]
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        const int q = 2 * kc + lk;
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const int ra = wm + 32 * i + lm;
            offa[i][kc] = (ra >> 4) * CHUNK + (ra & 15) * 64 + ((q ^ ((ra >> 2) & 3)) * 16);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rb = wn + 32 * j + lm;
            offb[j][kc] = XPL + (rb >> 4) * CHUNK + (rb & 15) * 64 + ((q ^ ((rb >> 2) & 3)) * 16);
        }
    }
    issue(s_begin, 0);
    for (int step = s_begin; step < s_end; ++step) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's DMA of step `step` has landed
        __syncthreads();                                       // everybody's has; everybody is done with the other stage
        if (step + 1 < s_end) issue(step + 1, (step + 1 - s_begin) & 1);
        const unsigned char* base = lds + ((step - s_begin) & 1) * STAGE;
        if (live == 0) continue;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 a[FA];
#pragma unroll
            for (int i = 0; i < FA; ++i) a[i] = *reinterpret_cast<const bf16x8*>(base + offa[i][kc]);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                bf16x8 b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(base + p * APL + offb[j][kc]);
#pragma unroll
                for (int i = 0; i < FA; ++i) {
                    if (i < live) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            // (split-k form, operands exchanged: the accumulator holds the TRANSPOSED tile, a lane's 32 neighbours run along m)
                            if constexpr (SPLITK) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    if constexpr (SPLITK) {
        // acc[i][j][r] is C[m][n] with m = 32i + (lane & 31), n = 32j + 8(r>>2) + 4(lane>>5) + (r&3); stored [n][m]
        float* const Cp = G.C + (long)split * G.part_stride;
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const int m = m0 + wm + 32 * i + lm;
            if (m >= G.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wn + 32 * j + 8 * (r >> 2) + 4 * lk + (r & 3);
                    if (n < G.N) Cp[(long)n * G.ldc + m] = acc[i][j][r];
                }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + lm;
            if (n >= G.N) continue;
            const float bias = G.bias ? G.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * lk + (r & 3);
                if (m >= G.M) continue;
                G.C[(long)m * G.ldc + n] = act_forward(ACT_OF(G, n), acc[i][j][r] + bias, ADD_OF(G, n));
            }
        }
}

static void launch_xgemm(XGemmArgs X, hipStream_t stream) {
    X.rows_fastest = 3L * X.N > (long)X.M ? 1 : 0;      // which operand is the larger one (both have Kp columns)
    // (measured, cfg 4's two products: 256-row tiles 66 us each at two waves per SIMD — fewer LDS reads per MFMA, but one
    //  workgroup per CU at these shapes — against 45 us with 128-row tiles at three; BSVI_XGEMM_TALL=1 selects the tall tile)
    static const bool tall_env = [] { const char* e = getenv("BSVI_XGEMM_TALL"); return e && e[0] == '1'; }();
    const bool tall = tall_env && X.M >= 256;
    static const bool glds = [] { const char* e = getenv("BSVI_XGEMM_GLDS"); return !(e && e[0] == '0'); }();
    if (glds) {
        const unsigned tiles_g = (unsigned)(((X.M + (tall ? 255 : 127)) / (tall ? 256 : 128)) * ((X.N + 127) / 128));
        if (tall) hipLaunchKernelGGL((xgemm_nt_glds_kernel<256, false>), dim3(tiles_g), dim3(256), 0, stream, X);
        else hipLaunchKernelGGL((xgemm_nt_glds_kernel<128, false>), dim3(tiles_g), dim3(256), 0, stream, X);
        return;
    }
    const int tbm = tall ? 256 : 128;
    const unsigned tiles = (unsigned)(((X.M + tbm - 1) / tbm) * ((X.N + 127) / 128));
    if (tall) hipLaunchKernelGGL((xgemm_nt_kernel<256>), dim3(tiles), dim3(256), 0, stream, X);
    else hipLaunchKernelGGL((xgemm_nt_kernel<128>), dim3(tiles), dim3(256), 0, stream, X);
}

// ---- the weight gradient of the data layer on the same pieces ---------------------------------------------------------
// dW[n][p] = sum_r dY[r][n] x[idx[r]][p]: the EXACT operand is x again and the f32 operand is dY, so dY is split into
// three bf16 pieces.  The MFMA wants k (= the row r) contiguous in both operands: dy_split_t_kernel writes the pieces
// transposed, [3][n_out][Rp], and xt_gather_kernel the gathered data rows transposed, [P][Rp] (it only needs the
// minibatch indices, so it runs on the side stream beside the forward pass).  xgemm_nt_glds_kernel<128, true> then
// multiplies slices of the rows (split-k) and writes one partial [n_out][P] per slice for reduce_partials.
// Rp = R rounded up to 64 (rows beyond R are zero in both operands).
struct XdwPlan { int Rp, steps, steps_per_split, slices, grid_splits, tiles; };
static XdwPlan xdw_plan(int P, int n_out, size_t R) {
    XdwPlan p{};
    p.Rp = (int)((R + 63) / 64 * 64);
    p.steps = p.Rp / XBK;
    p.tiles = ((P + 127) / 128) * ((n_out + 127) / 128);
    const int want = std::max(1, std::min(p.steps, 512 / std::max(p.tiles, 1)));     // two workgroups per CU, one round
    p.steps_per_split = (p.steps + want - 1) / want;
    p.slices = (p.steps + p.steps_per_split - 1) / p.steps_per_split;
    p.grid_splits = p.slices;
    while ((p.grid_splits * p.tiles) % 8) ++p.grid_splits;       // whole XCD shares; the padding workgroups exit
    return p;
}

// ---- f32 x f32 products on the bf16 matrix cores: six products of exact pieces ("bf16x6") ------------------------------------
// Every f32 is EXACTLY the sum of three bf16 numbers hi + mid + lo (8 + 8 + 8 significant bits, xw_split_kernel above), and a
// product of two bf16 numbers is exact in the f32 accumulator.  a * b = sum_ij a_i b_j over nine piece products; the three
// smallest (mid lo, lo mid, lo lo: <= 2^-24 |a b| each) are dropped, the other six — (hi hi), (hi mid), (mid hi), (hi lo), (lo hi),
// (mid mid) — run as six v_mfma_f32_32x32x16_bf16 per tile and k chunk: 6 / 16 of the f32-input MFMA's time (the bf16 one has 16x
// its rate), with the rounding behaviour of an f32 fma chain to within the dropped terms (~1.5 ulp of each product before
// accumulation, of random sign).  The WEIGHT operand's pieces are split once per iteration (x6_split_kernel: both orientations
// of every wide layer in one launch), so the kernel stages them with plain 16-byte copies; the ACTIVATION operand is read as
// f32 and split on the way into LDS (11 VALU instructions per pair of values, packed converts).
//   forward        C[M][N] = act(A[M][K] W[N][K]^T + bias)                      Bp = pieces of W   [3][N][Kp]
//   input gradient C[M][N] = (dY[M][K] W[K][N]) * act'(Y[M][N])  (+= C)        Bp = pieces of W^T [3][N][Kp]
struct X6Args {
    const float* A; int lda;                               // [M][K] f32: rows 16-byte aligned, K a multiple of 4
    const uint16_t* Bp; long plane_stride; int Kp;         // [3][N][Kp] bf16 pieces, zero beyond K; Kp a multiple of 32
    float* C; int ldc;
    int M, N, K;
    const float* bias;
    const float* Y; int ldy;
    int act; float post_add; int split; int act2; float post_add2;
    int accumulate;
    // round 6 — the Bernoulli likelihood FUSED into the product that makes the logits (forward, the decoder's last layer): the tile's
    // epilogue reads the data rows (their exact bf16 copy), stores d f / d logits where the logits would have gone and one partial
    // log-likelihood per (row, column tile) — the logits never reach memory and amort_lik is not launched (x6_epilogue, LIK)
    const uint16_t* lik_x; int lik_kp; const int32_t* lik_idx; float* lik_part;      // dataset_bf16 [DS][kp], row of every m, [tiles_n][M]
};

typedef __bf16 x6_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x6_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t x6_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t x6_pack(float a, float b) {       // (v_cvt_pk_bf16_f32: round to nearest even)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x6_f32x2{a, b}, x6_bf16x2));
}
// two f32 -> their pieces as packed pairs; hi + mid + lo == the value exactly
__device__ __forceinline__ void x6_split2(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = x6_pack(a, b);
    const float a1 = a - __uint_as_float(hi << 16), b1 = b - __uint_as_float(hi & 0xFFFF0000u);
    mid = x6_pack(a1, b1);
    lo = x6_pack(a1 - __uint_as_float(mid << 16), b1 - __uint_as_float(mid & 0xFFFF0000u));
}

struct X6SplitTable {
    struct Entry { const float* W; uint16_t* dst; int N, K, Kp, transposed; uint32_t first_block; } e[8];
    int n;
};
// pieces [3][N][Kp] of W [N][K] (transposed = 0) or of the transpose of W [K][N] (transposed = 1), zero beyond K: every wide
// layer's two orientations in ONE launch at the start of an iteration
__device__ __forceinline__ void x6_split_body(const X6SplitTable& T, uint32_t block) {
    int s = 0;
    while (s + 1 < T.n && block >= T.e[s + 1].first_block) ++s;
    const X6SplitTable::Entry E = T.e[s];
    const long i = (long)(block - E.first_block) * 256 + threadIdx.x;
    if (i >= (long)E.N * E.Kp) return;
    const int n = (int)(i / E.Kp), k = (int)(i - (long)n * E.Kp);
    const float w = k < E.K ? (E.transposed ? E.W[(long)k * E.N + n] : E.W[(long)n * E.K + k]) : 0.0f;
    const uint16_t hi = bf16_bits(w);
    const float r1 = w - bf16_value(hi);
    const uint16_t mid = bf16_bits(r1);
    const long plane = (long)E.N * E.Kp;
    E.dst[i] = hi; E.dst[plane + i] = mid; E.dst[2 * plane + i] = bf16_bits(r1 - bf16_value(mid));
}
__global__ __launch_bounds__(256) void x6_split_kernel(const X6SplitTable T) { x6_split_body(T, blockIdx.x); }

// Workgroup tile 128 x 128 (four waves of 64 x 64), k step 32, one LDS stage (six piece planes of 128 rows x 80 bytes: 60 KB,
// two workgroups per CU) with the next step's global loads in flight behind the MFMAs.  (Measured and dropped: the f32 operand's
// loads TWO steps ahead in a second register set — 240 VGPRs, the same times to the microsecond on every cfg 5 shape: the
// kernel does not wait for its loads.  profiles/r4/x6_notes.txt has what it does wait for.)
template <bool NN, int DBG = 0>      // DBG (BSVI_X6_DEBUG, timing only, wrong results): 1 no split, 2 no stores, 3 no MFMAs, 4 no LDS fragment reads
__global__ __launch_bounds__(256, 2) void x6gemm_r5_kernel(const X6Args G) {
    constexpr int TBM = 128, FA = 2, APL = TBM * XLD;
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * APL + 3 * XPLANE];      // A hi | mid | lo | B hi | mid | lo
    const int tiles_n = (G.N + 127) / 128;
    int bid = blockIdx.x;
    const int n_blocks = gridDim.x;
    if ((n_blocks & 7) == 0) bid = (bid & 7) * (n_blocks >> 3) + (bid >> 3);      // an XCD takes a contiguous range of tiles
    const int m0 = (bid / tiles_n) * TBM, n0 = (bid % tiles_n) * 128;              // (neighbours share the rows of A)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, lm = lane & 31, lk = lane >> 5;

    // staging.  A: 128 rows x 8 quads of f32 per step, four per thread (eight threads read 128 contiguous bytes of a row);
    // B: 3 x 128 x 4 pieces of 16 bytes, six per thread
    const float* pa[4];
    int sa[4], ka[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, row = c >> 3, kq = c & 7;
        pa[i] = G.A + (long)min(m0 + row, G.M - 1) * G.lda + kq * 4;
        sa[i] = row * XLD + kq * 8;
        ka[i] = kq * 4;
    }
    const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(G.Bp);
    uint32_t pb[6];
    int sb[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = tid + 256 * i, plane = c >> 9, w = c & 511, col = w >> 2, kq = w & 3;
        pb[i] = (uint32_t)((plane * G.plane_stride + (long)min(n0 + col, G.N - 1) * G.Kp + kq * 8) * 2);
        sb[i] = 3 * APL + plane * XPLANE + col * XLD + kq * 16;
    }
    const int n_steps = G.Kp / XBK;
    // (the last step of a K that is not a multiple of 32 reads quads beyond the row: those come from a valid address and are zeroed
    //  WHEN THEY ARE SPLIT.  Round 5: the select used to sit right behind the load — and the wave waited there, `s_waitcnt vmcnt`
    //  in FRONT of the step's MFMAs, for the loads it had just requested: the whole L2 round trip exposed in every step.)
    auto lda4 = [&](int i, int step) {
        const bool ok = step * XBK + ka[i] < G.K;
        return *reinterpret_cast<const f32x4*>(ok ? pa[i] + step * XBK : pa[i]);
    };
    auto ldw = [&](uint32_t off, int step) { return *reinterpret_cast<const xu4*>(wbase + off + (size_t)step * (XBK * 2)); };

    f32x16 acc[FA][2];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 ra[4];
    xu4 rb[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = lda4(i, 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) rb[i] = ldw(pb[i], 0);
    const unsigned char* at = lds + (wm + lm) * XLD + lk * 16;
    const unsigned char* bt = lds + 3 * APL + (wn + lm) * XLD + lk * 16;
    for (int step = 0; step < n_steps; ++step) {
        __syncthreads();                                   // the last step's reads of the stage are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t h0, m0_, l0, h1, m1, l1;
            const f32x4 v = (step * XBK + ka[i] < G.K) ? ra[i] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (DBG == 1) {
                h0 = m0_ = l0 = __float_as_uint(v[0]); h1 = m1 = l1 = __float_as_uint(v[2]);
            } else {
                x6_split2(v[0], v[1], h0, m0_, l0);
                x6_split2(v[2], v[3], h1, m1, l1);
            }
            *reinterpret_cast<x6_u2*>(lds + sa[i]) = x6_u2{h0, h1};
            *reinterpret_cast<x6_u2*>(lds + APL + sa[i]) = x6_u2{m0_, m1};
            *reinterpret_cast<x6_u2*>(lds + 2 * APL + sa[i]) = x6_u2{l0, l1};
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<xu4*>(lds + sb[i]) = rb[i];
        __syncthreads();
        if (step + 1 < n_steps) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = lda4(i, step + 1);
#pragma unroll
            for (int i = 0; i < 6; ++i) rb[i] = ldw(pb[i], step + 1);
        }
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 a[3][FA], b[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < FA; ++i) a[p][i] = *reinterpret_cast<const bf16x8*>(at + (DBG == 4 ? 0 : p * APL + 32 * i * XLD + kc * 32));
#pragma unroll
                for (int j = 0; j < 2; ++j) b[p][j] = *reinterpret_cast<const bf16x8*>(bt + (DBG == 4 ? 0 : p * XPLANE + 32 * j * XLD + kc * 32));
            }
            if (DBG == 3) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int i = 0; i < 2; ++i) { acc[i][0][p] += (float)a[p][i][0]; acc[i][1][p] += (float)b[p][i][0]; }
                continue;
            }
            // smallest products first: (lo hi), (hi lo), (mid mid), (mid hi), (hi mid), (hi hi)
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < FA; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], b[PB[q]][j], acc[i][j], 0, 0, 0);
        }
    }
    // epilogue: acc[i][j][r] is C[m][n] with m = 32i + 8(r>>2) + 4(lane>>5) + (r&3), n = 32j + (lane&31)   (as gemm_kernel)
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + lm;
            if (n >= G.N) continue;
            const float bias = (!NN && G.bias) ? G.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * lk + (r & 3);
                if (m >= G.M) continue;
                float v = acc[i][j][r];
                float* c = G.C + (long)m * G.ldc + n;
                if (DBG == 2) { if (v == 1.2345f) *c = v; continue; }
                if (!NN) {
                    *c = act_forward(ACT_OF(G, n), v + bias, ADD_OF(G, n));
                } else {
                    if (G.Y) v *= act_derivative(ACT_OF(G, n), G.Y[(long)m * G.ldy + n], ADD_OF(G, n));
                    *c = G.accumulate ? *c + v : v;
                }
            }
        }
}

// diagnostic build only (BSVI_X6_DEBUG=9): where a tile's time goes — s_memtime of one lane per workgroup at the start, at the top of the
// first, second and last step, behind the loop and behind the epilogue; the wall clock (100 MHz) at the start; the hardware id.  No
// product path reads or writes this; bsvi_debug_gemm mode 8 copies it out.
__device__ unsigned long long x6_stamps[8 * 8192];
__device__ __forceinline__ void x6_stamp(int slot) {
    if (threadIdx.x == 0 && blockIdx.x < 8192) x6_stamps[8 * blockIdx.x + slot] = __builtin_amdgcn_s_memtime();
}
// The tile epilogue of x6gemm_kernel: acc[j][r] is C[m][n] with m = 8(r>>2) + 4(lane>>5) + (r&3), n = 32j + (lane&31) of the wave's 32 x 128
// tile.  Through LDS (the stages are dead by then), one 32 x 64 half at a time (row stride 64 words: the 32 lanes of a ds_write_b32 group write
// one row, the 16 lanes of a ds_read_b128 group read 16 different bank quads of two rows), then 256 contiguous bytes of a row of C per
// 16 lanes: 16 stores of 16 bytes per thread instead of 64 of 4, bias / activation / derivative applied on the way.
template <bool NN, int DBG, bool LIK = false>
__device__ __forceinline__ void x6_epilogue(const X6Args& G, unsigned char* lds, const f32x16 (&acc)[4], int m0, int n0, int wave, int lane) {
    const int wm = wave * 32, lm = lane & 31, lk = lane >> 5;
    const int er = lane >> 4, ec = (lane & 15) * 4;
    // LIK: the data rows behind this lane's eight rows of the tile — requested before the barrier, and the data of a half before the
    // accumulators go through LDS, so that the two dependent round trips (row index, then the row's 8 bytes) hide behind them
    long lik_src[8];
    if (LIK) {
#pragma unroll
        for (int t = 0; t < 8; ++t) lik_src[t] = (long)G.lik_idx[min(m0 + wm + 4 * t + er, G.M - 1)] * G.lik_kp;
    }
    __syncthreads();                                           // every wave is done with the stages
    float* const tile = reinterpret_cast<float*>(lds) + wave * (32 * 64);
    float lik_row[8];                                         // LIK: the log-likelihood of rows 4t + er over this tile's columns
#pragma unroll
    for (int t = 0; t < 8; ++t) lik_row[t] = 0.0f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int n = n0 + 64 * half + ec;
        uint2 lik_xv[8];
        if (LIK) {
#pragma unroll
            for (int t = 0; t < 8; ++t) lik_xv[t] = *reinterpret_cast<const uint2*>(G.lik_x + lik_src[t] + min(n, G.N - 4));
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[(8 * (r >> 2) + 4 * lk + (r & 3)) * 64 + 32 * jj + lm] = acc[2 * half + jj][r];
        f32x4 bias4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (!NN && G.bias && n < G.N) {                        // (N is a multiple of 4; a bias may start anywhere in the parameter vector: scalar loads)
#pragma unroll
            for (int e = 0; e < 4; ++e) bias4[e] = G.bias[n + e];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = 4 * t + er, m = m0 + wm + row;
            f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * 64 + ec);
            if (LIK) {
                // x ~ Bernoulli(logits): log p = x l - (max(l, 0) + log(1 + exp(-|l|))), d log p / d l = x - sigmoid(l) — amort_lik's
                // arithmetic (the hardware exp2 / log2 / rcp forms), per element; the row's 64 columns of this half are summed over the
                // 16 lanes that hold them (a fixed butterfly), the two halves in order
                float lp = 0.0f;
                if (m < G.M && n < G.N) {
                    const uint2 xv = lik_xv[t];
                    const float xs[4] = {__uint_as_float(xv.x << 16), __uint_as_float(xv.x & 0xFFFF0000u), __uint_as_float(xv.y << 16), __uint_as_float(xv.y & 0xFFFF0000u)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lj = v[e] + bias4[e];
                        const float ex = __expf(-fabsf(lj)), one_e = 1.0f + ex, rc = __builtin_amdgcn_rcpf(one_e);
                        lp += xs[e] * lj - (fmaxf(lj, 0.0f) + __logf(one_e));
                        v[e] = xs[e] - (lj >= 0.0f ? rc : ex * rc);
                    }
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) lp += __shfl_xor(lp, o, 64);
                lik_row[t] += lp;
                if (m >= G.M || n >= G.N) continue;
                *reinterpret_cast<f32x4*>(G.C + (long)m * G.ldc + n) = v;
                continue;
            }
            if (m >= G.M || n >= G.N) continue;
            float* c = G.C + (long)m * G.ldc + n;
            if (DBG == 2) { if (v[0] == 1.2345f) *c = v[0]; continue; }
            if (!NN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_forward(ACT_OF(G, n + e), v[e] + bias4[e], ADD_OF(G, n + e));
            } else {
                if (G.Y) {
                    const f32x4 y = *reinterpret_cast<const f32x4*>(G.Y + (long)m * G.ldy + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= act_derivative(ACT_OF(G, n + e), y[e], ADD_OF(G, n + e));
                }
                if (G.accumulate) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = c0[e] + v[e];
                }
            }
            *reinterpret_cast<f32x4*>(c) = v;      // (non-temporal stores: single products 0-3 % faster, cfg 5 unchanged — profiles/r6/x6_nt.txt)
        }
    }
    if (LIK && (lane & 15) == 0) {
        float* const part = G.lik_part + (long)(n0 >> 7) * G.M;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int m = m0 + wm + 4 * t + er;
            if (m < G.M) part[m] = lik_row[t];
        }
    }
}
// Round 6: the same product with NOTHING staged through registers.  What bounded the kernel above was never the matrix pipe (63 % of
// its time remained with the MFMAs compiled out, profiles/r4/x6_notes.txt): per k step a workgroup pushed 48 KB through the CU's LDS
// STORE path (ds_write_b64 / b128 at ~79 B/clk/CU, with 2-way bank conflicts on both operands: 3.1 M SQ_LDS_BANK_CONFLICT cycles per
// launch, profiles/r5/cfg5_pmc_sq.csv) behind ~100 vector instructions of splitting, two barriers per step, and ended in 64 scalar
// 4-byte stores per thread.  Now:
//   * BOTH operands enter LDS by LDS-DMA (`global_load_lds_dwordx4`, the staging of xgemm_nt_glds_kernel): the weights' pieces as
//     before, the activations AS F32 — 16 KB per step instead of 24 KB of pieces, no vector instruction, no ds_write, two stages,
//     ONE barrier per step.  A wave's DMA writes 64 x 16 bytes linearly, so the conflict-free image is made on the SOURCE side: an
//     A chunk is 8 rows x 128 bytes and slot s of row r holds the row's 16-byte piece s ^ ((r >> 1) & 7) (the 16 lanes of a
//     ds_read_b128 group then read 16 different bank quads); the pieces' chunks keep xgemm_nt_glds_kernel's involution.
//   * the split of the activations happens in REGISTERS, on the fragment a wave is about to multiply (two ds_read_b128 of f32 ->
//     hi | mid | lo, 44 vector instructions per fragment) — in the shadow of the MFMAs (an MFMA holds the issue port for 8 of its
//     32 cycles).  A wave's tile is 32 rows x 128 columns, so every row of A is split by exactly one wave (64 x 64 wave tiles
//     would split it twice) and a wave's DMA fetches its own rows.
//   * the epilogue goes through LDS (the stages are dead by then): a wave writes its 32 x 64 half tile as the accumulators hold
//     it and reads rows back, 16 lanes x 16 bytes = 256 contiguous bytes of a row of C per quarter instruction — 16 stores of 16
//     bytes per thread instead of 64 of 4, bias / activation / derivative applied on the way.
// Same products in the same order as the kernel above: bit-identical results.  80 KB of LDS: two workgroups per CU.
template <bool NN, int DBG = 0, int VAR = 0, bool LIK = false>      // DBG (BSVI_X6_DEBUG, timing only, wrong results): 2 no stores, 3 no MFMAs, 4 fragment reads from one address, 5 no DMA after the first
__global__ __launch_bounds__(256, 2) void x6gemm_kernel(const X6Args G) {      // VAR (BSVI_X6_VAR): 0 the compiler's order of the main loop, 1 the hand-ordered loop
    constexpr int CHUNK = 1024, A_ST = 16 * CHUNK, B_PL = 8 * CHUNK, STAGE = A_ST + 3 * B_PL;      // f32 A (128 x 32) | hi | mid | lo (128 x 32 bf16 each)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
    const int tiles_n = (G.N + 127) / 128;
    int bid = blockIdx.x;
    const int n_blocks = gridDim.x;
    if ((n_blocks & 7) == 0) bid = (bid & 7) * (n_blocks >> 3) + (bid >> 3);      // an XCD takes a contiguous range of tiles
    const int m0 = (bid / tiles_n) * 128, n0 = (bid % tiles_n) * 128;              // (neighbours share the rows of A)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave * 32, lm = lane & 31, lk = lane >> 5;
    const int n_steps = G.Kp / XBK;
    const bool k_tail = (G.K % XBK) != 0;

    // DMA sources (`buffer_load_dwordx4 ... lds`: ONE 32-bit lane offset per piece, the k step in the SCALAR offset — no vector
    // arithmetic per load; the descriptors end with the matrices, so the pieces of the last row beyond K — the k tail of a K that is
    // not a multiple of 32 — are out of range: zeros, no access.  Other rows' tails read the next row's head; every fragment
    // beyond K is zeroed when it is split).
    // A: wave w fills chunks 4w .. 4w+3 = its own 32 rows; lane l of a chunk is slot (row l >> 3, piece l & 7).
    // The pieces: chunks 6w .. 6w+5 of the 24 (plane, 16 columns); lane l is slot (column l >> 2, piece l & 3).
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.A), 0, (int)(((size_t)(G.M - 1) * G.lda + G.K) * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(G.Bp), 0, (int)(3 * (size_t)G.plane_stride * 2), 0x00020000);
    uint32_t srca[4], srcb[6], dstb[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wm + 8 * i + (lane >> 3);
        const int gp = (lane & 7) ^ ((row >> 1) & 7);
        srca[i] = (uint32_t)(((size_t)min(m0 + row, G.M - 1) * G.lda + gp * 4) * sizeof(float));
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = 6 * wave + i, plane = c >> 3, chunk = c & 7, col = chunk * 16 + (lane >> 2);
        const int gp = (lane & 3) ^ ((lane >> 4) & 3);
        srcb[i] = (uint32_t)((plane * G.plane_stride + (long)min(n0 + col, G.N - 1) * G.Kp + gp * 8) * 2);
        dstb[i] = A_ST + plane * B_PL + chunk * CHUNK;
    }
    auto issue = [&](int step, int stage) {
        unsigned char* base = lds + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(base + (4 * wave + i) * CHUNK), 16, srca[i], step * (XBK * (int)sizeof(float)), 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_ptr_t)(base + dstb[i]), 16, srcb[i], step * (XBK * 2), 0, 0);
    };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // fragment addresses.  A: row R = wm + lm, the f32 k range [16 kc + 8 lk, + 8) = pieces q = 4 kc + 2 lk, q + 1 of the row:
    //   (R >> 3) * 1 KB + (R & 7) * 128 + ((q ^ ((R >> 1) & 7)) * 16).  The pieces of B: as xgemm_nt_glds_kernel.
    int offa[2][2], offb[4][2];
    {
        const int R = wm + lm, fa = (R >> 1) & 7;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
            for (int h = 0; h < 2; ++h) offa[kc][h] = (R >> 3) * CHUNK + (R & 7) * 128 + (((4 * kc + 2 * lk + h) ^ fa) * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = 32 * j + lm;
                offb[j][kc] = A_ST + (rb >> 4) * CHUNK + (rb & 15) * 64 + (((2 * kc + lk) ^ ((rb >> 2) & 3)) * 16);
            }
        }
    }
    if constexpr (VAR == 1) {
        // The hand-ordered loop.  hipcc leaves the 44 vector instructions of a fragment's split behind the chunk's last MFMA and waits
        // for every fragment read right in front of the MFMA that needs it (neither sched_group_barrier nor a software-pipelined
        // source changes that: the ISA is in profiles/r6/x6_notes.txt), so one wave never keeps the matrix pipe fed — measured: with
        // NO DMA at all the kernel loses 10 % of its time, with no MFMAs 35 %.  Here the order is written out and pinned
        // (`sched_barrier(0)` behind every MFMA): a chunk's twelve fragment reads of B and the two of the NEXT fragment of A go first,
        // then 24 MFMAs with, behind each, either one of the step's ten DMAs or one piece of the next fragment's split (twelve pieces
        // of 5 / 5 / 1 instructions per pair of values).  A wave fetches its OWN rows of A — the first four DMAs of a step — so it
        // reads the next step's first fragment from the other stage behind its own `vmcnt(6)`, no barrier, half a chunk before the
        // step ends.
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // smallest products first
        float x[8];
        uint32_t ph[4], pm[4], pl[4];
        auto take = [&](const f32x4& v0, const f32x4& v1, int k0, bool last) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = v0[e]; x[4 + e] = v1[e]; }
            if (last) {                                        // (K is a multiple of 4: a quad lies inside or outside)
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] = k0 < G.K ? x[e] : 0.0f; x[4 + e] = k0 + 4 < G.K ? x[4 + e] : 0.0f; }
            }
        };
        auto piece = [&](int c) {                              // piece c of the split of x[0..8): the arithmetic of x6_split2
            const int p = c & 3;
            float& u = x[2 * p];
            float& v = x[2 * p + 1];
            if (c < 4) { ph[p] = x6_pack(u, v); u -= __uint_as_float(ph[p] << 16); v -= __uint_as_float(ph[p] & 0xFFFF0000u); }
            else if (c < 8) { pm[p] = x6_pack(u, v); u -= __uint_as_float(pm[p] << 16); v -= __uint_as_float(pm[p] & 0xFFFF0000u); }
            else pl[p] = x6_pack(u, v);
        };
        auto pieces = [&](bf16x8 (&a)[3]) {
            a[0] = __builtin_bit_cast(bf16x8, xu4{ph[0], ph[1], ph[2], ph[3]});
            a[1] = __builtin_bit_cast(bf16x8, xu4{pm[0], pm[1], pm[2], pm[3]});
            a[2] = __builtin_bit_cast(bf16x8, xu4{pl[0], pl[1], pl[2], pl[3]});
        };
        auto dma = [&](int i, int step, unsigned char* base) {
            if (i < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(base + (4 * wave + i) * CHUNK), 16, srca[i], step * (XBK * (int)sizeof(float)), 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_ptr_t)(base + dstb[i - 4]), 16, srcb[i - 4], step * (XBK * 2), 0, 0);
        };
        bf16x8 a0[3], a1[3];
        if (DBG == 9) {
            x6_stamp(0);
            if (threadIdx.x == 0 && blockIdx.x < 8192) {
                x6_stamps[8 * blockIdx.x + 6] = __builtin_amdgcn_s_memrealtime();
                x6_stamps[8 * blockIdx.x + 7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
            }
        }
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // this wave's own rows of A (the first four DMAs) have landed
        {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(lds + offa[0][0]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(lds + offa[0][1]);
            take(v0, v1, 8 * lk, k_tail && n_steps == 1);
#pragma unroll
            for (int c = 0; c < 12; ++c) piece(c);
            pieces(a0);
        }
        auto body = [&](int step, auto more_t, auto last_t) {
            constexpr bool more = decltype(more_t)::value, last = decltype(last_t)::value;      // a step follows; this step holds the k tail
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of step `step` has landed
            __syncthreads();                                   // everybody's has; everybody is done with the other stage
            if (DBG == 9) { if (step == 0) x6_stamp(1); if (step == 1) x6_stamp(2); if (!more) x6_stamp(3); }
            const unsigned char* base = lds + (step & 1) * STAGE;
            unsigned char* nbase = lds + ((step + 1) & 1) * STAGE;
            bf16x8 b[3][4];
            // ---- chunk 0: a0 x the pieces of B; beside it the step's DMAs and the split of the chunk-1 fragment into a1
#pragma unroll
            for (int j = 0; j < 4; ++j) b[0][j] = *reinterpret_cast<const bf16x8*>(base + offb[j][0]);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(base + offa[1][0]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(base + offa[1][1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[2][j] = *reinterpret_cast<const bf16x8*>(base + 2 * B_PL + offb[j][0]);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[1][j] = *reinterpret_cast<const bf16x8*>(base + B_PL + offb[j][0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[PA[i >> 2]], b[PB[i >> 2]][i & 3], acc[i & 3], 0, 0, 0);
                if (i == 0) take(v0, v1, step * XBK + 16 + 8 * lk, last);
                if ((i & 1) == 0) { if (more && (i >> 1) < 10) dma(i >> 1, step + 1, nbase); }
                else piece(i >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            pieces(a1);
            // ---- chunk 1: a1 x the pieces of B; in its second half the next step's first fragment (this wave's own rows: vmcnt(6)) into a0
#pragma unroll
            for (int j = 0; j < 4; ++j) b[0][j] = *reinterpret_cast<const bf16x8*>(base + offb[j][1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[2][j] = *reinterpret_cast<const bf16x8*>(base + 2 * B_PL + offb[j][1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[1][j] = *reinterpret_cast<const bf16x8*>(base + B_PL + offb[j][1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[PA[i >> 2]], b[PB[i >> 2]][i & 3], acc[i & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) {
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // (the four DMAs of A were issued first)
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(nbase + offa[0][0]);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(nbase + offa[0][1]);
                __builtin_amdgcn_sched_barrier(0);
                take(w0, w1, (step + 1) * XBK + 8 * lk, k_tail && step + 2 == n_steps);
            }
#pragma unroll
            for (int i = 12; i < 24; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[PA[i >> 2]], b[PB[i >> 2]][i & 3], acc[i & 3], 0, 0, 0);
                if (more) piece(i - 12);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) pieces(a0);
        };
        for (int step = 0; step + 1 < n_steps; ++step) body(step, std::true_type{}, std::false_type{});
        if (k_tail) body(n_steps - 1, std::false_type{}, std::true_type{});
        else body(n_steps - 1, std::false_type{}, std::false_type{});
        if (DBG == 9) x6_stamp(4);
    } else {
        issue(0, 0);
        for (int step = 0; step < n_steps; ++step) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's DMA of step `step` has landed
            __syncthreads();                                       // everybody's has; everybody is done with the other stage
            if (step + 1 < n_steps && DBG != 5) issue(step + 1, (step + 1) & 1);
            const unsigned char* base = lds + (step & 1) * STAGE;
            const bool last = k_tail && step == n_steps - 1;
    #pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                f32x4 v0 = *reinterpret_cast<const f32x4*>(base + (DBG == 4 ? 0 : offa[kc][0]));
                f32x4 v1 = *reinterpret_cast<const f32x4*>(base + (DBG == 4 ? 0 : offa[kc][1]));
                if (last) {                                        // (K is a multiple of 4: a quad lies inside or outside)
                    const int k0 = step * XBK + 16 * kc + 8 * lk;
                    if (k0 >= G.K) v0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    if (k0 + 4 >= G.K) v1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
                uint32_t h[4], m[4], l[4];
                x6_split2(v0[0], v0[1], h[0], m[0], l[0]);
                x6_split2(v0[2], v0[3], h[1], m[1], l[1]);
                x6_split2(v1[0], v1[1], h[2], m[2], l[2]);
                x6_split2(v1[2], v1[3], h[3], m[3], l[3]);
                bf16x8 a[3];
                a[0] = __builtin_bit_cast(bf16x8, xu4{h[0], h[1], h[2], h[3]});
                a[1] = __builtin_bit_cast(bf16x8, xu4{m[0], m[1], m[2], m[3]});
                a[2] = __builtin_bit_cast(bf16x8, xu4{l[0], l[1], l[2], l[3]});
                bf16x8 b[3][4];
    #pragma unroll
                for (int p = 0; p < 3; ++p)
    #pragma unroll
                    for (int j = 0; j < 4; ++j) b[p][j] = *reinterpret_cast<const bf16x8*>(base + (DBG == 4 ? 0 : p * B_PL + offb[j][kc]));
                if (DBG == 3) {
    #pragma unroll
                    for (int p = 0; p < 3; ++p)
    #pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j][p] += (float)a[p][0] + (float)b[p][j][0];
                    continue;
                }
                // smallest products first: (lo hi), (hi lo), (mid mid), (mid hi), (hi mid), (hi hi)
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        #pragma unroll
                for (int q = 0; q < 6; ++q)
    #pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]], b[PB[q]][j], acc[j], 0, 0, 0);
                }
        }
    }
    x6_epilogue<NN, DBG, LIK>(G, lds, acc, m0, n0, wave, lane);
    if (DBG == 9) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); x6_stamp(5); }
}

// The weight gradient of a wide layer, C[M][N] = sum_r A[r][m] B[r][n] (A = dY, B = the layer's input: BOTH f32 activations), as six
// products of exact bf16 pieces (x6gemm_kernel's arithmetic) with the transposing split of both operands on the way into LDS — no
// transposed copy in memory (its traffic would cost what the product saves).  Workgroup = one 128 x 128 output tile x one slice of
// the rows (split k, a partial per slice, reduce_partials adds the slices in order); k step = 32 rows.  A thread owns ONE column
// of the A tile and one of the B tile for 16 of the 32 rows: 16 coalesced 4-byte loads each (a wave reads 256 contiguous bytes of
// a row), held one step ahead in registers; eight consecutive rows of a column are one 16-byte LDS store per piece, [column][k] —
// the layout the MFMA fragments read.  The column sums of A (the bias gradient) ride along in the n = 0 tiles.
struct X6TnArgs {
    const float* A; int lda;            // [K][M]
    const float* B; int ldb;            // [K][N]
    float* C; long part_stride;         // partial slices [slice][M][N]
    float* bias_grad;                   // [slice][M] or null
    int M, N, K, chunk, tiles;          // rows per slice (a multiple of 32), output tiles
    const int32_t* rows;                // gather of B's rows (the minibatch rows of the dataset), or null
};

// BX: B is exactly bf16 (the data rows): one piece, three products (a_lo b, a_mid b, a_hi b)
template <int DBG = 0, bool BX = false>
__global__ __launch_bounds__(256, 2) void x6tn_kernel(const X6TnArgs G) {
    constexpr int APL = XPLANE;
    __shared__ __attribute__((aligned(16))) unsigned char lds[6 * XPLANE];      // A hi | mid | lo | B hi | mid | lo, each [128][XLD]
    __shared__ float colsum[2][128];
    const int tiles_n = (G.N + 127) / 128;
    int bid = blockIdx.x;
    const int n_blocks = gridDim.x;
    if ((n_blocks & 7) == 0) bid = (bid & 7) * (n_blocks >> 3) + (bid >> 3);      // an XCD takes a contiguous range: the tiles of a slice share its rows
    const int slice = bid / G.tiles, tile = bid - slice * G.tiles;
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
    const int r_begin = slice * G.chunk, r_end = min(G.K, r_begin + G.chunk);
    if (r_begin >= G.K) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, lm = lane & 31, lk = lane >> 5;
    const int col = tid & 127, kg = tid >> 7;                                      // column of both tiles, rows 16 kg .. 16 kg + 15 of a step
    // buffer loads: ONE 32-bit lane offset per operand (the thread's column), the row in the scalar offset — no vector arithmetic per
    // load (with 64-bit lane addresses the 32 loads of a step cost ~100 vector instructions of address computation)
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.A), 0, -1, 0x00020000);
    // (B from the dataset's f32 rows also when a bf16 copy exists: 2-byte buffer loads made the gather twice as slow — 208 against 100 us)
    const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.B), 0, -1, 0x00020000);
    const uint32_t va_off = (uint32_t)min(m0 + col, G.M - 1) * 4u, vb_off = (uint32_t)min(n0 + col, G.N - 1) * 4u;
    const uint32_t b_row_bytes = (uint32_t)G.ldb * 4u;
    const int kgs = __builtin_amdgcn_readfirstlane(kg);                          // (a wave's threads share kg: rows as scalars)
    const int n_steps = (r_end - r_begin + XBK - 1) / XBK;
    float ra[16], rb[16];
    // gathered rows (BX): the indices of a step are loaded a step ahead of the loads they address (a dependent pair of round trips
    // — index, then a random dataset row from HBM — otherwise sits in front of every step), as VECTOR loads of one address (scalar
    // loads share the LDS counter: every s_load in the product phase would turn its fine-grained LDS waits into lgkmcnt(0))
    const __amdgpu_buffer_rsrc_t rows_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(G.rows ? G.rows : reinterpret_cast<const int32_t*>(G.A)), 0, -1, 0x00020000);
    uint32_t ridx[16];
    auto fetch_idx = [&](int step) {
        if (!BX) return;
        const int r0 = r_begin + step * XBK + 16 * kgs;
#pragma unroll
        for (int i = 0; i < 16; ++i) ridx[i] = __builtin_amdgcn_raw_buffer_load_b32(rows_rsrc, 0u, (uint32_t)min(r0 + i, r_end - 1) * 4u, 0);
    };
    auto fetch = [&](int step) {
        const int r0 = r_begin + step * XBK + 16 * kgs;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + i;
            const bool ok = r < r_end;
            const uint32_t rr = (uint32_t)(ok ? r : r_end - 1);
            const float va = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra_rsrc, va_off, rr * (uint32_t)G.lda * 4u, 0));
            float vb;
            if (BX) {
                vb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb_rsrc, ridx[i] * b_row_bytes + vb_off, 0u, 0));
            } else {
                vb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb_rsrc, vb_off, rr * b_row_bytes, 0));
            }
            // (rows beyond the slice are zeroed WHEN THEY ARE SPLIT, mask(): a select right behind the load is where the wave waits for
            //  it — in front of the step's MFMAs, with the whole round trip exposed)
            ra[i] = va;
            rb[i] = vb;
        }
    };
    auto mask = [&](int step) {
        const int valid = r_end - (r_begin + step * XBK + 16 * kgs);      // (the same for a wave's threads)
        if (valid >= 16) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ra[i] = i < valid ? ra[i] : 0.0f;
            rb[i] = i < valid ? rb[i] : 0.0f;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float asum = 0.0f;
    fetch_idx(0);
    fetch(0);
    if (n_steps > 1) fetch_idx(1);
    const unsigned char* at = lds + (wm + lm) * XLD + lk * 16;
    const unsigned char* bt = lds + 3 * APL + (wn + lm) * XLD + lk * 16;
    unsigned char* const sa = lds + col * XLD + kg * 32;                           // two 16-byte stores per piece: rows 16 kg .. +7, +8 .. +15
    unsigned char* const sb = lds + 3 * APL + col * XLD + kg * 32;
    for (int step = 0; step < n_steps; ++step) {
        __syncthreads();                                   // the last step's reads of the stage are done
        mask(step);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t hi[4], mid[4], lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x6_split2(ra[8 * h + 2 * q], ra[8 * h + 2 * q + 1], hi[q], mid[q], lo[q]);
            *reinterpret_cast<xu4*>(sa + 16 * h) = xu4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<xu4*>(sa + APL + 16 * h) = xu4{mid[0], mid[1], mid[2], mid[3]};
            *reinterpret_cast<xu4*>(sa + 2 * APL + 16 * h) = xu4{lo[0], lo[1], lo[2], lo[3]};
            if (BX) {
#pragma unroll
                for (int q = 0; q < 4; ++q) hi[q] = x6_pack(rb[8 * h + 2 * q], rb[8 * h + 2 * q + 1]);
                *reinterpret_cast<xu4*>(sb + 16 * h) = xu4{hi[0], hi[1], hi[2], hi[3]};
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) x6_split2(rb[8 * h + 2 * q], rb[8 * h + 2 * q + 1], hi[q], mid[q], lo[q]);
                *reinterpret_cast<xu4*>(sb + 16 * h) = xu4{hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<xu4*>(sb + APL + 16 * h) = xu4{mid[0], mid[1], mid[2], mid[3]};
                *reinterpret_cast<xu4*>(sb + 2 * APL + 16 * h) = xu4{lo[0], lo[1], lo[2], lo[3]};
            }
        }
        if (G.bias_grad) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asum += ra[i];     // (rows in order; the two halves of a step are joined below)
        }
        __syncthreads();
        if (step + 1 < n_steps) {
            fetch(step + 1);                                 // (its indices arrived during the last step)
            if (step + 2 < n_steps) fetch_idx(step + 2);
        }
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 a[3][2], b[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[p][i] = *reinterpret_cast<const bf16x8*>(at + p * APL + 32 * i * XLD + kc * 32);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[p][j] = *reinterpret_cast<const bf16x8*>(bt + ((BX && p) ? 0 : p * APL) + 32 * j * XLD + kc * 32);
            }
            if (DBG == 3) continue;
            // smallest products first: (lo hi), (hi lo), (mid mid), (mid hi), (hi mid), (hi hi); exact B: (lo b), (mid b), (hi b)
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            constexpr int PAX[3] = {2, 1, 0};
#pragma unroll
            for (int q = 0; q < (BX ? 3 : 6); ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[BX ? PAX[q] : PA[q]][i], b[BX ? 0 : PB[q]][j], acc[i][j], 0, 0, 0);
        }
    }
    // the slice's partial: acc[i][j][r] is C[m][n] with m = 32i + 8(r>>2) + 4(lane>>5) + (r&3), n = 32j + (lane&31)
    float* const part = G.C + (long)slice * G.part_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + lm;
            if (n >= G.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * lk + (r & 3);
                if (m < G.M) part[(long)m * G.N + n] = acc[i][j][r];
            }
        }
    if (G.bias_grad && n0 == 0) {
        colsum[kg][col] = asum;
        __syncthreads();
        if (kg == 0 && m0 + col < G.M) G.bias_grad[(long)slice * G.M + m0 + col] = colsum[0][col] + colsum[1][col];
    }
}

struct X6TnPlan { int tiles, splits, chunk, slices; };
// 128 x 128 tiles, two workgroups per CU: as many slices of the rows as fill 512 slots in ONE round
static X6TnPlan x6tn_plan(int M, int N, int K) {
    X6TnPlan p{};
    p.tiles = ((M + 127) / 128) * ((N + 127) / 128);
    static const int slots = [] { const char* e = getenv("BSVI_X6TN_SLOTS"); return e ? atoi(e) : 512; }();
    const int splits = std::max(1, std::min((K + XBK - 1) / XBK, slots / std::max(p.tiles, 1)));
    p.chunk = ((K + splits - 1) / splits + XBK - 1) / XBK * XBK;
    p.slices = (K + p.chunk - 1) / p.chunk;
    p.splits = p.slices;
    return p;
}

// dY [R][ld] f32 -> T [3][N][Rp] bf16 pieces, and the column sums of every 64-row block (the bias gradient's partials)
__global__ __launch_bounds__(256) void dy_split_t_kernel(const float* dY, int ld, int R, int Rp, int N, uint16_t* T, float* colsum) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, n0 = blockIdx.y * 64, t = threadIdx.x;
    {
        const int c = t & 63, n = n0 + c;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (t >> 6) + 4 * i, r = r0 + row;
            tile[row][c] = (r < R && n < N) ? dY[(long)r * ld + n] : 0.0f;
        }
    }
    __syncthreads();
    const int c = t >> 2, g = t & 3, n = n0 + c;
    uint16_t piece[3][16];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float w = tile[16 * g + i][c];
        s += w;
        const uint16_t hi = bf16_bits(w);
        const float r1 = w - bf16_value(hi);
        const uint16_t mid = bf16_bits(r1);
        piece[0][i] = hi; piece[1][i] = mid; piece[2][i] = bf16_bits(r1 - bf16_value(mid));
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (n >= N) return;
    if (colsum && g == 0) colsum[(long)blockIdx.x * N + n] = s;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        xu4 lo, hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lo[q] = (uint32_t)piece[p][2 * q] | ((uint32_t)piece[p][2 * q + 1] << 16);
            hi[q] = (uint32_t)piece[p][8 + 2 * q] | ((uint32_t)piece[p][9 + 2 * q] << 16);
        }
        xu4* out = reinterpret_cast<xu4*>(T + ((long)p * N + n) * Rp + r0 + 16 * g);
        out[0] = lo; out[1] = hi;
    }
}

// Xb [DS][Kp] bf16, idx [R] -> XT [P][Rp] bf16: XT[p][r] = Xb[idx[r]][p].  A workgroup transposes 64 rows x 128 columns:
// 16-byte loads along the rows into LDS (row stride 65 words), then a thread takes a PAIR of columns (one word) over eight
// rows and regroups the halves into the two 16-byte column pieces; the eight threads of a column pair are neighbours,
// so a column's 64 rows leave as one 128-byte line.
__global__ __launch_bounds__(256) void xt_gather_kernel(const uint16_t* Xb, const int32_t* idx, int Kp, int P, int R, int Rp, uint16_t* XT) {
    __shared__ uint32_t tile[64][65];
    const int r0 = blockIdx.x * 64, p0 = blockIdx.y * 128, t = threadIdx.x;
    {
        const int piece = t & 15;              // 16-byte piece of the 256-byte row segment
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (t >> 4) + 16 * i, r = r0 + row;
            xu4 w = {0u, 0u, 0u, 0u};
            if (r < R && p0 + 8 * piece < Kp) w = *reinterpret_cast<const xu4*>(Xb + (long)idx[r] * Kp + p0 + 8 * piece);
#pragma unroll
            for (int q = 0; q < 4; ++q) tile[row][4 * piece + q] = w[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int item = t + 256 * k, g = item & 7, q = item >> 3, p = p0 + 2 * q;
        if (p >= P) continue;
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = tile[8 * g + i][q];
        xu4 even, odd;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            even[h] = (w[2 * h] & 0xFFFFu) | (w[2 * h + 1] << 16);
            odd[h] = (w[2 * h] >> 16) | (w[2 * h + 1] & 0xFFFF0000u);
        }
        *reinterpret_cast<xu4*>(XT + (long)p * Rp + r0 + 8 * g) = even;
        if (p + 1 < P) *reinterpret_cast<xu4*>(XT + (long)(p + 1) * Rp + r0 + 8 * g) = odd;
    }
}

// partial[slice][n_out][P] = (dY^T x) over the slice's rows.  XT, T as above.
static void launch_xdw(const uint16_t* XT, const uint16_t* T, int P, int n_out, const XdwPlan& plan, float* partial, hipStream_t stream) {
    XGemmArgs X{};
    X.X = XT; X.rows = nullptr; X.Wp = T; X.plane_stride = (long)n_out * plan.Rp;
    X.C = partial; X.ldc = P; X.M = P; X.N = n_out; X.Kp = plan.Rp;
    X.steps_per_split = plan.steps_per_split; X.part_stride = (long)n_out * P;
    X.rows_fastest = 1;
    hipLaunchKernelGGL((xgemm_nt_glds_kernel<128, true>), dim3((unsigned)(plan.tiles * plan.grid_splits)), dim3(256), 0, stream, X);
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- layers with a side of width <= 8 (latent heads, first decoder layer): memory-bound, no matrix cores ----------
constexpr int SKINNY = 8;

__device__ __forceinline__ float skinny_epilogue(const GemmArgs& G, int mode, float v, long r, int n) {
    float* c = G.C + r * G.ldc + n;
    if (mode == MODE_NT) return act_forward(ACT_OF(G, n), v + (G.bias ? G.bias[n] : 0.0f), ADD_OF(G, n));
    if (G.Y) v *= act_derivative(ACT_OF(G, n), G.Y[r * G.ldy + n], ADD_OF(G, n));
    return G.accumulate ? *c + v : v;
}

// K <= 8:  C[r][n] = epilogue(sum_k A[r][k] * B(k, n)),  B(k, n) = B[k * sbk + n * sbn].  One thread per output.
__global__ __launch_bounds__(256) void skinny_k_kernel(const GemmArgs G, int sbk, int sbn, int mode) {
    // (32-bit index arithmetic: launch_gemm routes here only when M * N fits; a 64-bit divide costs more than the K <= 8 FMAs)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint32_t)G.M * (uint32_t)G.N) return;
    const uint32_t r = i / (uint32_t)G.N;
    const int n = (int)(i - r * (uint32_t)G.N);
    const float* a = G.A + (G.rows ? (long)G.rows[r] : (long)r) * G.lda;
    float acc = 0.0f;
    for (int k = 0; k < G.K; ++k) acc += a[k] * G.B[(long)k * sbk + (long)n * sbn];
    G.C[(long)r * G.ldc + n] = skinny_epilogue(G, mode, acc, (long)r, n);
}

// the same for 16-byte-aligned operands and N % 4 == 0: four adjacent outputs per thread, vector loads of B (when it
// is stored [K][N]), of the bias / of Y / of C, and one 16-byte store
__global__ __launch_bounds__(256) void skinny_k4_kernel(const GemmArgs G, int sbk, int sbn, int mode) {
    const uint32_t quads = (uint32_t)G.N >> 2;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint32_t)G.M * quads) return;
    const uint32_t r32 = i / quads;
    const int n = (int)(i - r32 * quads) * 4;
    const long r = (long)r32;
    const float* a = G.A + (G.rows ? (long)G.rows[r] : r) * G.lda;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < G.K; ++k) {
        const float av = a[k];
        float b[4];
        if (sbn == 1) {
            const float4 bv = *reinterpret_cast<const float4*>(G.B + (long)k * sbk + n);
            b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = G.B[(long)k * sbk + (long)(n + j) * sbn];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += av * b[j];
    }
    float* c = G.C + r * G.ldc + n;
    float out[4];
    if (mode == MODE_NT) {
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (G.bias) bias = *reinterpret_cast<const float4*>(G.bias + n);
        const float bb[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = act_forward(ACT_OF(G, n + j), acc[j] + bb[j], ADD_OF(G, n + j));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = acc[j];
        if (G.Y) {
            const float4 y = *reinterpret_cast<const float4*>(G.Y + r * G.ldy + n);
            const float yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] *= act_derivative(ACT_OF(G, n + j), yy[j], ADD_OF(G, n + j));
        }
        if (G.accumulate) {
            const float4 old = *reinterpret_cast<const float4*>(c);
            out[0] += old.x; out[1] += old.y; out[2] += old.z; out[3] += old.w;
        }
    }
    *reinterpret_cast<float4*>(c) = make_float4(out[0], out[1], out[2], out[3]);
}

// N <= 8:  row dot products.  B is staged transposed in LDS ([n][k]); one wave per row, lanes stride over k.
__global__ __launch_bounds__(256) void skinny_n_kernel(const GemmArgs G, int sbk, int sbn, int mode, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float bt[];   // [N][K]
    for (int i = threadIdx.x; i < G.N * G.K; i += 256) {
        const int n = i / G.K, k = i - n * G.K;
        bt[i] = G.B[(long)k * sbk + (long)n * sbn];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = G.vecA && (G.K & 3) == 0;
    const long r_end = min((long)G.M, ((long)blockIdx.x + 1) * rows_per_block);
    long r = (long)blockIdx.x * rows_per_block + wave;
    if (vec) {
        // RP rows per wave and pass: all their loads are in flight before the first is reduced
        constexpr int RP = 2;
        for (; r + 4 * (RP - 1) < r_end; r += 4 * RP) {
            const float* a[RP];
            float acc[RP][SKINNY];
#pragma unroll
            for (int q = 0; q < RP; ++q) {
                const long rq = r + 4 * q;
                a[q] = G.A + (G.rows ? (long)G.rows[rq] : rq) * G.lda;
#pragma unroll
                for (int n = 0; n < SKINNY; ++n) acc[q][n] = 0.0f;
            }
            for (int k = lane * 4; k < G.K; k += 256) {
                float4 av[RP];
#pragma unroll
                for (int q = 0; q < RP; ++q) av[q] = *reinterpret_cast<const float4*>(a[q] + k);
#pragma unroll
                for (int n = 0; n < SKINNY; ++n)
                    if (n < G.N) {
                        const float4 bv = *reinterpret_cast<const float4*>(&bt[n * G.K + k]);
#pragma unroll
                        for (int q = 0; q < RP; ++q)
                            acc[q][n] += av[q].x * bv.x + av[q].y * bv.y + av[q].z * bv.z + av[q].w * bv.w;
                    }
            }
#pragma unroll
            for (int n = 0; n < SKINNY; ++n) {
                if (n >= G.N) break;
#pragma unroll
                for (int q = 0; q < RP; ++q) {
                    const float v = wave_sum64(acc[q][n]);
                    if (lane == n) G.C[(r + 4 * q) * G.ldc + n] = skinny_epilogue(G, mode, v, r + 4 * q, n);
                }
            }
        }
    }
    for (; r < r_end; r += 4) {
        const float* a = G.A + (G.rows ? (long)G.rows[r] : r) * G.lda;
        float acc[SKINNY];
#pragma unroll
        for (int n = 0; n < SKINNY; ++n) acc[n] = 0.0f;
        if (vec) {
            for (int k = lane * 4; k < G.K; k += 256) {
                const float4 av = *reinterpret_cast<const float4*>(a + k);
#pragma unroll
                for (int n = 0; n < SKINNY; ++n)
                    if (n < G.N) {
                        const float4 bv = *reinterpret_cast<const float4*>(&bt[n * G.K + k]);
                        acc[n] += av.x * bv.x + av.y * bv.y + av.z * bv.z + av.w * bv.w;
                    }
            }
        } else {
            for (int k = lane; k < G.K; k += 64) {
                const float av = a[k];
#pragma unroll
                for (int n = 0; n < SKINNY; ++n)
                    if (n < G.N) acc[n] += av * bt[n * G.K + k];
            }
        }
#pragma unroll
        for (int n = 0; n < SKINNY; ++n) {
            if (n >= G.N) break;
            const float v = wave_sum64(acc[n]);
            if (lane == n) G.C[r * G.ldc + n] = skinny_epilogue(G, mode, v, r, n);
        }
    }
}

// weight gradient with a narrow side: out(w, j) += sum_r Wide[r][w] * Narrow[r][j], j < narrow <= 8, one thread per w
// and a chunk of rows per workgroup.  narrow_is_a: A (dY) is the narrow operand -> out = C[j][w], else out = C[w][j].
// bias_grad: column sums of A (dY) on the side.
__global__ __launch_bounds__(256) void skinny_tn_kernel(const GemmArgs G, int narrow_is_a, int rows_per_block) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    const int wide = narrow_is_a ? G.N : G.M, narrow = narrow_is_a ? G.M : G.N;
    const float* Wd = narrow_is_a ? G.B : G.A;
    const float* Nr = narrow_is_a ? G.A : G.B;
    const int ldw = narrow_is_a ? G.ldb : G.lda, ldn = narrow_is_a ? G.lda : G.ldb;
    const long r0 = (long)blockIdx.y * rows_per_block, r1 = min((long)G.K, r0 + rows_per_block);
    float acc[SKINNY], nsum[SKINNY], wsum = 0.0f;
#pragma unroll
    for (int j = 0; j < SKINNY; ++j) acc[j] = nsum[j] = 0.0f;
    if (w < wide) {
#pragma unroll 4
        for (long r = r0; r < r1; ++r) {
            // the gathered operand (data rows) is always B
            const long rw = (!narrow_is_a || !G.rows) ? r : (long)G.rows[r];
            const long rn = (narrow_is_a || !G.rows) ? r : (long)G.rows[r];
            const float wv = Wd[rw * ldw + w];
            wsum += wv;
#pragma unroll
            for (int j = 0; j < SKINNY; ++j)
                if (j < narrow) {
                    const float nv = Nr[rn * ldn + j];
                    acc[j] += wv * nv;
                    nsum[j] += nv;
                }
        }
        float* part = G.C + (long)blockIdx.y * G.part_stride;            // partial of this chunk of rows
#pragma unroll
        for (int j = 0; j < SKINNY; ++j)
            if (j < narrow) part[narrow_is_a ? (long)j * G.ldc + w : (long)w * G.ldc + j] = acc[j];
        if (G.bias_grad) {
            float* bpart = G.bias_grad + (long)blockIdx.y * G.M;
            if (!narrow_is_a) {
                bpart[w] = wsum;
            } else if (w == 0) {
#pragma unroll
                for (int j = 0; j < SKINNY; ++j)
                    if (j < narrow) bpart[j] = nsum[j];
            }
        }
    }
}

// ---- the narrow layers, second generation ------------------------------------------------------------------------
// (the kernels above are the general fallbacks; at the shapes of BASELINE config 5 they ran at 1.2-1.9 TB/s: one wave per
//  row with a 64-lane butterfly per output, or one dword per thread and step)
template <int CTRL>
__device__ __forceinline__ float dpp16(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_total(float v) {      // every lane of a 16-lane DPP row gets the row's sum
    v += dpp16<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp16<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp16<0x141>(v);    // row_half_mirror
    v += dpp16<0x140>(v);    // row_mirror
    return v;
}

// N <= NP <= 8 row dot products, K % 4 == 0:  C[r][n] = epilogue(sum_k A[r][k] B(k, n)).  Sixteen lanes per row, four rows
// per wave and pass, two passes in flight: every lane has up to eight independent 16-byte loads outstanding, and the
// sum over a row's sixteen lanes is four DPP steps per output.  B is staged once per workgroup, transposed and padded to
// NP outputs with zeros, so that the inner loop has no conditions.
template <int NP>
__global__ __launch_bounds__(256) void rowdot_kernel(const GemmArgs G, int sbk, int sbn, int mode, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float bt[];   // [NP][K]
    const int K = G.K, K4 = K >> 2;
    for (int i = threadIdx.x; i < NP * K; i += 256) {
        const int n = i / K, k = i - n * K;
        bt[i] = n < G.N ? G.B[(long)k * sbk + (long)n * sbn] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, sub = lane >> 4;
    const long r_begin = (long)blockIdx.x * rows_per_block, r_end = min((long)G.M, r_begin + rows_per_block);
    constexpr int RP = 2;
    for (long r0 = r_begin + wave * 4 * RP; r0 < r_end; r0 += 16 * RP) {
        const float* a[RP];
        long row[RP];
        float acc[RP][NP];
#pragma unroll
        for (int q = 0; q < RP; ++q) {
            row[q] = r0 + 4 * q + sub;
            const long rr = min(row[q], (long)G.M - 1);                      // rows past the end re-read the last one
            a[q] = G.A + (G.rows ? (long)G.rows[rr] : rr) * G.lda;
#pragma unroll
            for (int n = 0; n < NP; ++n) acc[q][n] = 0.0f;
        }
        for (int k4 = l16; k4 < K4; k4 += 64) {                              // four 16-byte loads per row in flight
            f32x4 av[RP][4];
#pragma unroll
            for (int q = 0; q < RP; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = min(k4 + 16 * u, K4 - 1);
                    av[q][u] = *reinterpret_cast<const f32x4*>(a[q] + 4 * kk);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k4 + 16 * u;
                if (kk < K4) {
#pragma unroll
                    for (int n = 0; n < NP; ++n) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(&bt[n * K + 4 * kk]);
#pragma unroll
                        for (int q = 0; q < RP; ++q)
                            acc[q][n] += (av[q][u].x * bv.x + av[q][u].y * bv.y) + (av[q][u].z * bv.z + av[q][u].w * bv.w);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < RP; ++q) {
            float mine = 0.0f;
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const float v = row16_total(acc[q][n]);
                mine = (l16 == n) ? v : mine;
            }
            if (l16 < G.N && row[q] < r_end) G.C[row[q] * G.ldc + l16] = skinny_epilogue(G, mode, mine, row[q], l16);
        }
    }
}

// weight gradient with a narrow side (narrow <= NP <= 8, wide % 4 == 0), one partial per chunk of rows:
//   out(w, j) = sum_{r in chunk} Wide[r][w] * Narrow[r][j]
// A thread owns four adjacent w and every RS-th row of the chunk (RS = 256 / (wide / 4) row slots), eight rows in flight;
// the row slots are added in order through LDS.  Column sums of A ride along (bias gradient).
template <int NP>
__global__ __launch_bounds__(256) void outer_kernel(const GemmArgs G, int narrow_is_a, int rows_per_block) {
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    const int wide = narrow_is_a ? G.N : G.M, narrow = narrow_is_a ? G.M : G.N;
    const float* Wd = narrow_is_a ? G.B : G.A;
    const float* Nr = narrow_is_a ? G.A : G.B;
    const int ldw = narrow_is_a ? G.ldb : G.lda, ldn = narrow_is_a ? G.lda : G.ldb;
    const int quads = wide >> 2, q0 = blockIdx.x * 256;
    const int tpr = min(256, quads - q0), slots = 256 / tpr;                 // threads per row, row slots
    const int tw = threadIdx.x % tpr, slot = threadIdx.x / tpr;
    const bool active = slot < slots;
    const int w = (q0 + tw) * 4;
    const long r0 = (long)blockIdx.y * rows_per_block, r1 = min((long)G.K, r0 + rows_per_block);
    f32x4 acc[NP];
    f32x4 wsum = f32x4(0.0f);
    float nsum[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { acc[j] = f32x4(0.0f); nsum[j] = 0.0f; }
    if (active) {
        constexpr int U = 8;
        for (long r = r0 + slot; r < r1; r += (long)slots * U) {
            f32x4 wv[U];
            float nv[U][NP];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long rr = min(r + (long)u * slots, r1 - 1);            // past the end: re-read, weight 0 below
                // the gathered operand (data rows) is always B
                const long rw = (!narrow_is_a || !G.rows) ? rr : (long)G.rows[rr];
                const long rn = (narrow_is_a || !G.rows) ? rr : (long)G.rows[rr];
                wv[u] = *reinterpret_cast<const f32x4*>(Wd + rw * ldw + w);
                // (16-byte loads: columns past `narrow` are the padding of the value buffers — they only reach accumulators
                //  that are never stored)
#pragma unroll
                for (int j4 = 0; j4 < NP; j4 += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(Nr + rn * ldn + j4);
                    nv[u][j4] = v.x;
                    if (j4 + 1 < NP) nv[u][j4 + 1] = v.y;
                    if (j4 + 2 < NP) nv[u][j4 + 2] = v.z;
                    if (j4 + 3 < NP) nv[u][j4 + 3] = v.w;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r + (long)u * slots < r1) {
                    wsum += wv[u];
#pragma unroll
                    for (int j = 0; j < NP; ++j) { acc[j] += wv[u] * nv[u][j]; nsum[j] += nv[u][j]; }
                }
            }
        }
    }
    // row slots in order: slot 0 adds the others' values as they stand in LDS
    float* part = G.C + (long)blockIdx.y * G.part_stride;
    auto fold = [&](f32x4 v) -> f32x4 {
        __syncthreads();
        if (active) *reinterpret_cast<f32x4*>(&red[(slot * tpr + tw) * 4]) = v;
        __syncthreads();
        f32x4 t = f32x4(0.0f);
        if (slot == 0)
            for (int s2 = 0; s2 < slots; ++s2) t += *reinterpret_cast<const f32x4*>(&red[(s2 * tpr + tw) * 4]);
        return t;
    };
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (j >= narrow) break;
        const f32x4 t = fold(acc[j]);
        if (slot == 0) {
            if (narrow_is_a) {
                *reinterpret_cast<f32x4*>(&part[(long)j * G.ldc + w]) = t;
            } else {
                part[(long)(w + 0) * G.ldc + j] = t.x; part[(long)(w + 1) * G.ldc + j] = t.y;
                part[(long)(w + 2) * G.ldc + j] = t.z; part[(long)(w + 3) * G.ldc + j] = t.w;
            }
        }
    }
    if (G.bias_grad) {
        float* bpart = G.bias_grad + (long)blockIdx.y * G.M;
        if (!narrow_is_a) {
            const f32x4 t = fold(wsum);
            if (slot == 0) *reinterpret_cast<f32x4*>(&bpart[w]) = t;
        } else if (blockIdx.x == 0) {
            f32x4 lo = f32x4(0.0f), hi = f32x4(0.0f);
            lo.x = nsum[0]; if (NP > 1) lo.y = nsum[1]; if (NP > 2) lo.z = nsum[2]; if (NP > 3) lo.w = nsum[3];
            if (NP > 4) { hi.x = nsum[4]; hi.y = nsum[5]; hi.z = nsum[6]; hi.w = nsum[7]; }
            const f32x4 tl = fold(lo), th = NP > 4 ? fold(hi) : f32x4(0.0f);
            if (threadIdx.x == 0) {
                const float t[8] = {tl.x, tl.y, tl.z, tl.w, th.x, th.y, th.z, th.w};
                for (int j = 0; j < narrow; ++j) bpart[j] = t[j];
            }
        }
    }
}

// K <= 8 (compile time), forward layout (B stored [N][K], N % 4 == 0): four adjacent outputs per thread; their 4 K weights
// are contiguous.  One thread = 1 + K 16-byte loads, one 16-byte store.
template <int K>
__global__ __launch_bounds__(256) void skinny_k4nt_kernel(const GemmArgs G) {
    const uint32_t quads = (uint32_t)G.N >> 2;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint32_t)G.M * quads) return;
    const uint32_t r32 = i / quads;
    const int n = (int)(i - r32 * quads) * 4;
    const long r = (long)r32;
    const float* a = G.A + (G.rows ? (long)G.rows[r] : r) * G.lda;
    float av[K], bl[4 * K];
#pragma unroll
    for (int k = 0; k < K; ++k) av[k] = a[k];
    const f32x4* bq = reinterpret_cast<const f32x4*>(G.B + (long)n * K);     // 4 K floats = K 16-byte pieces, aligned
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const f32x4 v = bq[k];
        bl[4 * k] = v.x; bl[4 * k + 1] = v.y; bl[4 * k + 2] = v.z; bl[4 * k + 3] = v.w;
    }
    const f32x4 bias = G.bias ? *reinterpret_cast<const f32x4*>(G.bias + n) : f32x4(0.0f);
    const float bb[4] = {bias.x, bias.y, bias.z, bias.w};
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) acc += av[k] * bl[j * K + k];
        out[j] = act_forward(ACT_OF(G, n + j), acc + bb[j], ADD_OF(G, n + j));
    }
    *reinterpret_cast<f32x4*>(G.C + r * G.ldc + n) = f32x4{out[0], out[1], out[2], out[3]};
}

// ---- fixed-order sums of the partial results ------------------------------------------------------------------------
// dst(m, n) = [dst(m, n) +] sum_s src[s * stride + m * cols + n], s ascending: the second stage of every reduction over
// rows (weight / bias gradients: one partial per k split; loss sums: one partial per workgroup of amort_latent_bwd).
constexpr int kMaxSegments = 40;
struct Segment {
    float* dst;
    const float* src;
    uint32_t rows, cols, ldd;     // dst is [rows][ldd], the partials are [splits][rows][cols]
    uint32_t splits;
    uint32_t first_block;         // of this segment in the launch
    uint32_t add;                 // keep what dst holds
    uint32_t vec;                 // 16-byte accesses (set by segment_blocks)
    uint32_t wide;                // more than 64 slices: a workgroup's four waves share them (set by segment_blocks)
};
struct SegmentTable {
    Segment seg[kMaxSegments];
    int n;
};

// eight partials in flight per thread; the association is fixed: slice s goes to accumulator s mod 8, the accumulators
// are added pairwise at the end
template <typename V>
__device__ __forceinline__ V sum_slices(const V* p, size_t stride, uint32_t splits) {
    V a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = V(0.0f);
    uint32_t s = 0;
    for (; s + 8 <= splits; s += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += p[(size_t)(s + j) * stride];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (s + j < splits) a[j] += p[(size_t)(s + j) * stride];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

__global__ __launch_bounds__(256) void reduce_partials(const SegmentTable T) {
    int k = 0;
    while (k + 1 < T.n && blockIdx.x >= T.seg[k + 1].first_block) ++k;
    const Segment S = T.seg[k];
    const uint32_t total = S.rows * S.cols;
    uint32_t i = (blockIdx.x - S.first_block) * 256u + threadIdx.x;
    if (S.wide) {
        // many slices, few outputs (the narrow layers: one slice per 64 rows): 64 outputs per workgroup, its four waves
        // take the slices w, w + 4, ... and are added in wave order through LDS
        __shared__ float quarter[4][64];
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        i = (blockIdx.x - S.first_block) * 64u + lane;
        float v = 0.0f;
        if (i < total) {
            const uint32_t mine = (S.splits + 3u - wave) / 4u;                       // slices wave, wave + 4, ...
            v = sum_slices(S.src + (size_t)wave * total + i, (size_t)4 * total, mine);
        }
        quarter[wave][lane] = v;
        __syncthreads();
        if (wave == 0 && i < total) {
            v = (quarter[0][lane] + quarter[1][lane]) + (quarter[2][lane] + quarter[3][lane]);
            const uint32_t m = i / S.cols, n = i - m * S.cols;
            float* d = S.dst + (size_t)m * S.ldd + n;
            *d = S.add ? *d + v : v;
        }
        return;
    }
    if (S.vec) {                       // four adjacent outputs per thread, 16-byte loads and stores
        i *= 4u;
        if (i >= total) return;
        const f32x4 v = sum_slices(reinterpret_cast<const f32x4*>(S.src + i), total / 4u, S.splits);
        const uint32_t m = i / S.cols, n = i - m * S.cols;
        f32x4* d = reinterpret_cast<f32x4*>(S.dst + (size_t)m * S.ldd + n);
        *d = S.add ? *d + v : v;
    } else {
        if (i >= total) return;
        const float v = sum_slices(S.src + i, (size_t)total, S.splits);
        const uint32_t m = i / S.cols, n = i - m * S.cols;
        float* d = S.dst + (size_t)m * S.ldd + n;
        *d = S.add ? *d + v : v;
    }
}

// blocks of reduce_partials a segment needs; decides whether it can use 16-byte accesses
static uint32_t segment_blocks(Segment& S) {
    const uint32_t total = S.rows * S.cols;
    S.wide = S.splits > 64 ? 1u : 0u;
    if (S.wide) { S.vec = 0; return (total + 63) / 64; }
    S.vec = (S.cols % 4 == 0 && S.ldd % 4 == 0 && (uintptr_t)S.src % 16 == 0 && (uintptr_t)S.dst % 16 == 0 && total % 4 == 0) ? 1u : 0u;
    return ((S.vec ? total / 4 : total) + 255) / 256;
}

// ---- row-wise pieces -------------------------------------------------------------------------------------------
struct RowParams {
    int R, B, DS, P, Dz;
    int n_local, sample_base;
    uint32_t seed_lo, seed_hi, off_lo, off_hi;
    int estimator;
    float entropy_const;     // log(n_samples_global): the reference's entropy of the minibatch variable
    const int32_t* indices_in;
    int32_t* idx;            // [R] dataset row of every (sample, batch) row
    int32_t* indices_out;
    const float* noise_in;
    float* noise_out;
    const float* dataset;    // [DS][P] device
    const uint16_t* dataset_bf16; int data_kp;     // its exact bf16 copy [DS][data_kp], or null
    const float* loc;  int ld_loc;  int act_loc;  float add_loc;
    const float* scale; int ld_scale; int act_scale; float add_scale;
    float* dloc; float* dscale;
    float* eps;              // [R][Dz]
    float* z; int ld_z;      // decoder input
    const float* dz;         // decoder-input gradient, [R][ld_z]
    const float* prior_loc; const float* prior_scale;
    // a learnable prior: raw values in the parameter buffer (loc as stored, scale = softplus(raw)); kNoOffset: the constants above
    const float* params; uint32_t prior_loc_off, prior_scale_off;
    float* prior_loc_part; float* prior_scale_part;      // [workgroups of amort_latent_bwd][Dz] partial gradient sums, or null
    int likelihood; const float* lik_scale;              // BSVI_AMORT_LIK_*; [P] scale of the Normal likelihood
    // ... or learnable: raw values in the parameter buffer (scale = softplus(raw); 1 of them, or P); kNoOffset: the constants above
    uint32_t lik_scale_off, lik_scale_size;
    float* lik_scale_part;                               // [row slices][P] partial sums of d log p / d raw, or null
    // ... or a decoder head: per row and feature (post-activation), its gradient buffer (pre-activation), the head's activation
    const float* lik_sd; float* dlik_sd; int ld_lik_sd, act_lik_sd; float add_lik_sd;
    float* rowf;             // [R] f per row
    const float* lik_part; int lik_tiles;              // (round 6) [lik_tiles][R] partial log-likelihoods of the fused likelihood epilogue (x6_epilogue), or null
    float* rowlq;            // [R] log q per row
    float* logits; int ld_logits;
    float* out;
    float* sum_part;         // [workgroups of amort_latent_bwd][2]: partial sums of the estimator value / non-finite count
    float* fvalue_out; float* logq_out;
    // caller-weighted gradients (bsvi_amort_args::f_weight_dev / q_weight_dev — the second pass of a user-defined gradient
    // estimator): row r's gradient seeds are a_r grad f_r + b_r grad log q_r; every launch behind the row kernels is linear
    // in the seeds, so the weights enter where the seeds are written (amort_lik, amort_latent_bwd) and nowhere else
    const float* f_weight; const float* q_weight;      // [R] each, or null (a = 1; BlackBox: b = f)
};

// minibatch of sample s: a keyed bijection of [0, DS) per (seed, iteration, sample) — 4-round Feistel on the next
// power of four, cycle-walked — evaluated at b = 0..B-1: B distinct rows, like np.random.choice(replace=False)
// (distributions.py:438-440)
__device__ __forceinline__ uint32_t minibatch_row(const RowParams& D, uint32_t s_global, uint32_t b) {
    uint32_t half_bits = 1;
    while ((1u << (2 * half_bits)) < (uint32_t)D.DS) ++half_bits;
    const uint32_t mask = (1u << half_bits) - 1u;
    uint32_t x = b;
    for (int walk = 0; walk < 64; ++walk) {
        uint32_t lft = (x >> half_bits) & mask, rgt = x & mask;
        for (uint32_t round = 0; round < 4; ++round) {
            const u32x4 h = philox4x32(rgt, round | (s_global << 2), D.off_lo, D.off_hi, D.seed_lo ^ 0x7f4a7c15u, D.seed_hi);
            const uint32_t t = lft ^ (h.x & mask);
            lft = rgt;
            rgt = t;
        }
        x = (lft << half_bits) | rgt;
        if (x < (uint32_t)D.DS) return x;
    }
    return b % (uint32_t)D.DS;
}

__device__ __forceinline__ void amort_rows_body(const RowParams& D, int r) {
    if (r >= D.R) return;
    const int s = r / D.B, b = r - s * D.B;
    const int32_t i = D.indices_in ? D.indices_in[r] : (int32_t)minibatch_row(D, (uint32_t)(D.sample_base + s), (uint32_t)b);
    D.idx[r] = i;
    if (D.indices_out) D.indices_out[r] = i;
}
__global__ void amort_rows(const RowParams D) { amort_rows_body(D, blockIdx.x * blockDim.x + threadIdx.x); }
// the heads of an iteration in ONE launch of heterogeneous workgroups: the minibatch rows, and behind them the bf16 pieces of
// every weight matrix the matrix-core products read (three launches of 6 - 8 us each, back to back on an idle chip, before)
__global__ __launch_bounds__(256) void amort_head(const RowParams D, const X6SplitTable T, uint32_t row_blocks) {
    if (blockIdx.x < row_blocks) amort_rows_body(D, blockIdx.x * 256 + threadIdx.x);
    else x6_split_body(T, blockIdx.x - row_blocks);
}

// p(z)'s parameters for latent dimension d: constants, or the learnable raw values behind the constructor's transforms
// (loc: identity; scale: 0 + softplus(raw), torch's softplus with threshold 20 — geometric_ranges.py RightHalfLine);
// dscale = d scale / d raw
constexpr uint32_t kNoOffset = 0xFFFFFFFFu;
__device__ __forceinline__ void prior_of(const RowParams& D, int d, float& loc, float& scale, float& dscale) {
    loc = D.prior_loc_off != kNoOffset ? D.params[D.prior_loc_off + d] : D.prior_loc[d];
    if (D.prior_scale_off != kNoOffset) {
        const float raw = D.params[D.prior_scale_off + d];
        scale = raw > 20.0f ? raw : log1pf(expf(raw));
        dscale = raw > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-raw));
    } else {
        scale = D.prior_scale[d];
        dscale = 0.0f;
    }
}

// z = loc + scale * eps; per row: log p(z), H[q(z|x)], log q(z|x)   (torch normal.py:83-116)
__global__ void amort_latent_fwd(const RowParams D) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= D.R) return;
    const uint32_t row_global = (uint32_t)(D.sample_base * D.B + r);
    float lp = 0.0f, H = 0.0f, lq = 0.0f;
    for (int d = 0; d < D.Dz; d += 2) {
        float e0, e1 = 0.0f;
        if (D.noise_in) {
            e0 = D.noise_in[(long)r * D.Dz + d];
            if (d + 1 < D.Dz) e1 = D.noise_in[(long)r * D.Dz + d + 1];
        } else {
            const u32x4 x = philox4x32(row_global, (uint32_t)(d >> 1), D.off_lo, D.off_hi, D.seed_lo, D.seed_hi);
            bsvi::box_muller(x.x, x.y, e0, e1);
        }
        for (int j = 0; j < 2 && d + j < D.Dz; ++j) {
            const int dd = d + j;
            const float e = j ? e1 : e0;
            const float m = D.loc[(long)r * D.ld_loc + dd], sd = D.scale[(long)r * D.ld_scale + dd];
            const float z = m + sd * e;
            D.eps[(long)r * D.Dz + dd] = e;
            if (D.noise_out) D.noise_out[(long)r * D.Dz + dd] = e;
            D.z[(long)r * D.ld_z + dd] = z;
            float pl, ps, dps;
            prior_of(D, dd, pl, ps, dps);
            const float u = (z - pl) / ps;
            const float lsd = logf(sd);
            lp += -0.5f * u * u - logf(ps) - kHalfLog2Pi;
            H += 0.5f + kHalfLog2Pi + lsd;
            lq += -0.5f * e * e - lsd - kHalfLog2Pi;
        }
    }
    D.rowf[r] = lp + H + D.entropy_const;
    D.rowlq[r] = lq;
}

// the Normal likelihood's scale for feature j: a constant, or softplus of its learnable raw value (one for all features, or one each)
__device__ __forceinline__ float lik_scale_of(const RowParams& D, int j) {
    if (D.lik_scale_off == kNoOffset) return D.lik_scale[j];
    const float raw = D.params[D.lik_scale_off + (D.lik_scale_size == 1u ? 0 : j)];
    return raw > 20.0f ? raw : log1pf(expf(raw));
}

// A learnable likelihood scale (ABI 10): d log p / d raw_j = dsoftplus(raw_j) * sum_rows (u^2 - 1) / s_j with u = (x - mean) / s_j.
// amort_lik left d log p / d mean = u / s_j where the means stood, so (u^2 - 1) / s = (u / s)^2 s - 1 / s needs no second pass over
// the data.  Thread j walks one slice of the rows (consecutive threads: consecutive columns of the row-major matrix); the slices
// are added by the iteration's one reduction launch in a fixed order.
__global__ __launch_bounds__(256) void amort_lik_scale_grad(const RowParams D, int rows_per_slice) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= D.P) return;
    const float s = lik_scale_of(D, j);
    const float raw = D.params[D.lik_scale_off + (D.lik_scale_size == 1u ? 0 : j)];
    const float ds = raw > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-raw));
    const int r0 = blockIdx.y * rows_per_slice, r1 = min(D.R, r0 + rows_per_slice);
    if (D.f_weight) {
        // amort_lik left a_r u / s: a_r (u^2 - 1) / s = g'^2 s / a_r - a_r / s   (a_r = 0: the row contributes nothing)
        float acc = 0.0f;
        for (int r = r0; r < r1; ++r) {
            const float a = D.f_weight[r], g = D.logits[(long)r * D.ld_logits + j];
            acc += a != 0.0f ? g * g * s / a - a / s : 0.0f;
        }
        D.lik_scale_part[(long)blockIdx.y * D.P + j] = ds * acc;
        return;
    }
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        const float g0 = D.logits[(long)r * D.ld_logits + j], g1 = D.logits[(long)(r + 1) * D.ld_logits + j];
        const float g2 = D.logits[(long)(r + 2) * D.ld_logits + j], g3 = D.logits[(long)(r + 3) * D.ld_logits + j];
        a0 += g0 * g0; a1 += g1 * g1; a2 += g2 * g2; a3 += g3 * g3;
    }
    for (; r < r1; ++r) { const float g = D.logits[(long)r * D.ld_logits + j]; a0 += g * g; }
    const float sum_g2 = (a0 + a1) + (a2 + a3);
    D.lik_scale_part[(long)blockIdx.y * D.P + j] = ds * (sum_g2 * s - (float)(r1 - r0) / s);
}

// log p(x | z) = sum_j x_j l_j - softplus(l_j)  (torch binomial.py:140-160 with total_count = 1) and
// dlogits = x - sigmoid(l), written over the logits.  One wave per row.
__global__ __launch_bounds__(256) void amort_lik(const RowParams D) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= D.R) return;
    const float* x = D.dataset + (long)D.idx[r] * D.P;
    float* l = D.logits + (long)r * D.ld_logits;
    float lp = 0.0f;
    const float aw = D.f_weight ? D.f_weight[r] : 1.0f;       // (weight of this row's gradient seeds)
    if (D.likelihood == 1) {
        // x ~ Normal(mean = decoder value, scale_j): log p = -u^2/2 - log(scale) - log(2 pi)/2, u = (x - mean)/scale;
        // d log p / d mean = u / scale, written over the means   (torch normal.py:83-90)
        if (D.lik_sd) {          // the scale is a second head of the decoder: d log p / d scale = (u^2 - 1) / scale, through the head's activation
            const float* sd = D.lik_sd + (long)r * D.ld_lik_sd;
            float* dsd = D.dlik_sd + (long)r * D.ld_lik_sd;
            for (int j = lane; j < D.P; j += 64) {
                const float sj = sd[j], u = (x[j] - l[j]) / sj;
                lp += -0.5f * u * u - logf(sj) - kHalfLog2Pi;
                l[j] = aw * (u / sj);
                dsd[j] = aw * ((u * u - 1.0f) / sj * act_derivative(D.act_lik_sd, sj, D.add_lik_sd));
            }
            lp = wave_sum64(lp);
            if (lane == 0) D.rowf[r] += lp;
            return;
        }
        for (int j = lane; j < D.P; j += 64) {
            const float sj = lik_scale_of(D, j), u = (x[j] - l[j]) / sj;
            lp += -0.5f * u * u - logf(sj) - kHalfLog2Pi;
            l[j] = aw * (u / sj);
        }
        lp = wave_sum64(lp);
        if (lane == 0) D.rowf[r] += lp;
        return;
    }
    auto element = [&](float lj, float xj) {
        // e in (0, 1]: log(1 + e) through the hardware log2 is good to ~1e-7 ABSOLUTE, which is what matters in a sum
        // whose other term is max(l, 0); the reciprocal to 1 ulp
        const float e = __expf(-fabsf(lj));
        const float one_e = 1.0f + e, r = __builtin_amdgcn_rcpf(one_e);
        lp += xj * lj - (fmaxf(lj, 0.0f) + __logf(one_e));
        const float sig = lj >= 0.0f ? r : e * r;
        return aw * (xj - sig);
    };
    if (D.dataset_bf16 && ((D.P | D.ld_logits) & 3) == 0) {      // the exact bf16 copy of the data: half the bytes of the gather
        const uint16_t* xb = D.dataset_bf16 + (long)D.idx[r] * D.data_kp;
        for (int q = lane; q < (D.P >> 2); q += 64) {
            const float4 lv = *reinterpret_cast<const float4*>(l + 4 * q);
            const uint2 xv = *reinterpret_cast<const uint2*>(xb + 4 * q);
            float4 o;
            o.x = element(lv.x, __uint_as_float(xv.x << 16)); o.y = element(lv.y, __uint_as_float(xv.x & 0xFFFF0000u));
            o.z = element(lv.z, __uint_as_float(xv.y << 16)); o.w = element(lv.w, __uint_as_float(xv.y & 0xFFFF0000u));
            *reinterpret_cast<float4*>(l + 4 * q) = o;
        }
    } else if (((D.P | D.ld_logits) & 3) == 0) {          // 16-byte accesses: a wave covers 1 KiB of the row per pass
        for (int q = lane; q < (D.P >> 2); q += 64) {
            const float4 lv = *reinterpret_cast<const float4*>(l + 4 * q), xv = *reinterpret_cast<const float4*>(x + 4 * q);
            float4 o;
            o.x = element(lv.x, xv.x); o.y = element(lv.y, xv.y); o.z = element(lv.z, xv.z); o.w = element(lv.w, xv.w);
            *reinterpret_cast<float4*>(l + 4 * q) = o;
        }
    } else {
        for (int j = lane; j < D.P; j += 64) l[j] = element(l[j], x[j]);
    }
    lp = wave_sum64(lp);
    if (lane == 0) D.rowf[r] += lp;
}

// joins the decoder's dz with the prior, entropy and (BlackBox) score-function terms and accumulates the loss sums.
//   pathwise:  value_r = f_r
//   BlackBox:  value_r = lq_r * stopgrad(f_r) + f_r;  d lq_r / d scale = -1/scale (z - loc = scale*eps cancels the rest)
__global__ __launch_bounds__(256) void amort_latent_bwd(const RowParams D) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float value = 0.0f, bad = 0.0f;
    if (r < D.R) {
        float f = D.rowf[r];
        if (D.lik_part) {                                  // the column tiles' partial log-likelihoods, in tile order
            float lp = 0.0f;
            for (int t = 0; t < D.lik_tiles; ++t) lp += D.lik_part[(long)t * D.R + r];
            f += lp;
        }
        const float lq = D.rowlq[r];
        value = D.estimator == 1 ? lq * f + f : f;
        if (!isfinite(value)) bad = 1.0f;
        if (D.fvalue_out) D.fvalue_out[r] = f;
        if (D.logq_out) D.logq_out[r] = lq;
        // (a caller's weights: a_r on every term of f_r — the decoder's dz carries it already —, b_r where stopgrad(f_r) stands)
        const float aw = D.f_weight ? D.f_weight[r] : 1.0f;
        const float score = D.q_weight ? D.q_weight[r] : (D.estimator == 1 ? f : 0.0f);
        for (int d = 0; d < D.Dz; ++d) {
            const float e = D.eps[(long)r * D.Dz + d];
            const float m = D.loc[(long)r * D.ld_loc + d], sd = D.scale[(long)r * D.ld_scale + d];
            const float z = D.z[(long)r * D.ld_z + d];
            float pl, ps, dps;
            prior_of(D, d, pl, ps, dps);
            const float gz = D.dz[(long)r * D.ld_z + d] - aw * ((z - pl) / (ps * ps));
            const float gsd = gz * e + (aw - score) / sd;
            D.dloc[(long)r * D.ld_loc + d] = gz * act_derivative(D.act_loc, m, D.add_loc);
            D.dscale[(long)r * D.ld_scale + d] = gsd * act_derivative(D.act_scale, sd, D.add_scale);
        }
    }
    if (D.prior_loc_part) {
        // a learnable prior: d f / d loc = u / scale, d f / d raw scale = (u^2 - 1) / scale * dscale with u = (z - loc) / scale,
        // summed over the workgroup's rows per latent dimension (f enters both estimators' values with weight one)
        __shared__ float pr[2][4];
        for (int d = 0; d < D.Dz; ++d) {
            float gl = 0.0f, gs = 0.0f;
            if (r < D.R) {
                float pl, ps, dps;
                prior_of(D, d, pl, ps, dps);
                const float u = (D.z[(long)r * D.ld_z + d] - pl) / ps;
                const float aw = D.f_weight ? D.f_weight[r] : 1.0f;
                gl = aw * (u / ps);
                gs = aw * ((u * u - 1.0f) / ps * dps);
            }
            gl = wave_sum64(gl);
            gs = wave_sum64(gs);
            if ((threadIdx.x & 63) == 0) { pr[0][threadIdx.x >> 6] = gl; pr[1][threadIdx.x >> 6] = gs; }
            __syncthreads();
            if (threadIdx.x < 2) {
                float* dst = threadIdx.x ? D.prior_scale_part : D.prior_loc_part;
                dst[(long)blockIdx.x * D.Dz + d] = (pr[threadIdx.x][0] + pr[threadIdx.x][1]) + (pr[threadIdx.x][2] + pr[threadIdx.x][3]);
            }
            __syncthreads();
        }
    }
    // workgroup partial of the two sums (waves in order): reduce_partials adds the workgroups in order
    __shared__ float red[2][4];
    value = wave_sum64(value);
    bad = wave_sum64(bad);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = value; red[1][threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x < 2)
        D.sum_part[blockIdx.x * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// dst[r][c] = src[r][c] for c < n: moves caller-layout rows into / out of the padded value buffers
__global__ void copy_rows(const float* src, int lds, float* dst, int ldd, long rows, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n) return;
    const long r = rows * n < (1l << 32) ? (long)((uint32_t)i / (uint32_t)n) : i / n;
    const int c = (int)(i - r * n);
    dst[r * ldd + c] = src[r * lds + c];
}

}  // namespace bsvi_amort_impl

// =============================================================================================================
//  C ABI
// =============================================================================================================
using namespace bsvi_amort_impl;

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return bsvi_fail(BSVI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

struct Net {
    std::vector<bsvi_mlp_layer> layers;
    std::vector<int> width;        // per value
    std::vector<int> producer;     // layer index producing the value (-1: the input)
    std::vector<int> ld;           // leading dimension of the value's buffers
    std::vector<size_t> val_off, grad_off;   // float offsets into the workspace, per row-block base (times R at use)
};

// the fork / join events between the two streams of one call order work on ONE device: no system-scope fence (the default
// writes the caches back so that the host could inspect the data: ~6 us of idle stream behind every record; BSVI_AMORT_EVENT_FENCE=1
// keeps it)
static unsigned event_flags() {
    static const unsigned f = [] {
        const char* e = getenv("BSVI_AMORT_EVENT_FENCE");
        return (unsigned)hipEventDisableTiming | ((e && e[0] == '1') ? 0u : (unsigned)hipEventDisableSystemFence);
    }();
    return f;
}
#define kEventFlags event_flags()
struct bsvi_amort {
    bsvi_amort_desc d;
    Net enc, dec;
    float* dataset_dev = nullptr;
    // every dataset value is exactly a bf16: the layers that read the data rows run on the bf16 matrix cores at f32
    // accuracy (xgemm_nt_kernel).  dataset_bf16_dev [DS][data_kp] (rows zero padded to a multiple of 32 columns);
    // weight_pieces[l]: [3][n_out][data_kp] bf16 pieces of encoder layer l's weights, refreshed every iteration
    bool data_exact = false;
    int data_kp = 0;
    uint16_t* dataset_bf16_dev = nullptr;
    std::vector<uint16_t*> weight_pieces;
    bool xdw = false;              // ... and their weight gradients too (dy_split_t_kernel, xt_gather_kernel, the split-k product)
    float* prior_dev = nullptr;    // [2][Dz]
    float* lik_scale_dev = nullptr;   // [P], Normal likelihood
    size_t floats_per_row = 0;     // workspace floats per row (values + gradients + per-row scalars)
    // the weight-gradient GEMM of a layer and its input-gradient GEMM are independent: the former runs on a side
    // stream (fork on an event per layer, one join before the function returns) and fills the CUs the latter's
    // last wave of workgroups leaves idle
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> ready;
    hipEvent_t joined = nullptr;
    bool overlap = true;
    bool layers_cover_params = false;   // every parameter is a weight or bias of some layer: reduce_partials writes the whole block
    // the wide f32 layers on the bf16 matrix cores as six products of exact pieces (x6gemm_kernel): per layer the pieces of the
    // weights in both orientations, [3][n_out][kp_nt] for the forward product and [3][n_in][kp_nn] for the input gradient,
    // refreshed by ONE launch at the start of every iteration (BSVI_AMORT_X6=0: the f32-input MFMA kernel serves them)
    struct X6Layer { uint16_t* nt = nullptr; uint16_t* nn = nullptr; int kp_nt = 0, kp_nn = 0; };
    std::vector<X6Layer> x6[2];         // [0] encoder, [1] decoder
    bool x6_any = false;
    // Several ranks (round 5): the backward pass finishes the DECODER's gradients first.  When the host names a bucket stream
    // (bsvi_amort_set_bucket_stream) the reduction of the decoder's partial sums is launched THERE as soon as the decoder's weight
    // gradients are in flight, and the host's all-reduce of that range of the output block — the decoder's parameters are one
    // contiguous range [bucket_first, bucket_first + bucket_count) of the parameter vector when this is set — follows it on that
    // stream, beside the encoder's backward pass; the rest of the block is reduced at the end as always.
    hipStream_t bucket_stream = nullptr;
    hipEvent_t bucket_ev[2] = {nullptr, nullptr};
    uint32_t bucket_first = 0, bucket_count = 0;
};

static int pad4(int n) { return (n + 3) / 4 * 4; }

static int build_net(Net& net, const bsvi_mlp_layer* layers, uint32_t n_layers, int input_width, uint32_t n_params) {
    net.layers.assign(layers, layers + n_layers);
    uint32_t n_values = 1;
    for (const auto& l : net.layers) n_values = std::max(n_values, std::max(l.in_value, l.out_value) + 1);
    net.width.assign(n_values, -1);
    net.producer.assign(n_values, -1);
    net.width[0] = input_width;
    for (size_t i = 0; i < net.layers.size(); ++i) {
        const auto& l = net.layers[i];
        if (l.out_value == 0 || net.width[l.out_value] != -1) return bsvi_fail(BSVI_ERR_INVALID, "a network value is produced twice");
        if (net.width[l.in_value] == -1) return bsvi_fail(BSVI_ERR_INVALID, "network layers are not in topological order");
        if ((int)l.n_in != net.width[l.in_value]) return bsvi_fail(BSVI_ERR_INVALID, "layer input width mismatch");
        if (l.activation > BSVI_ACT_SOFTPLUS || l.activation2 > BSVI_ACT_SOFTPLUS) return bsvi_fail(BSVI_ERR_UNSUPPORTED, "unknown activation");
        if ((size_t)l.weight_off + (size_t)l.n_in * l.n_out > n_params) return bsvi_fail(BSVI_ERR_INVALID, "layer weights exceed the parameter buffer");
        if (l.bias_off != 0xFFFFFFFFu && (size_t)l.bias_off + l.n_out > n_params) return bsvi_fail(BSVI_ERR_INVALID, "layer bias exceeds the parameter buffer");
        net.width[l.out_value] = (int)l.n_out;
        net.producer[l.out_value] = (int)i;
    }
    net.ld.resize(n_values);
    for (uint32_t v = 0; v < n_values; ++v) net.ld[v] = pad4(net.width[v]);
    return BSVI_OK;
}

extern "C" int bsvi_amort_create(const bsvi_amort_desc* desc, bsvi_amort** out) {
    if (!desc || !out) return bsvi_fail(BSVI_ERR_INVALID, "null argument");
    BSVI_CHECK_STRUCT(desc, bsvi_amort_desc);
    if (desc->abi_version != BSVI_ABI_VERSION) return bsvi_fail(BSVI_ERR_INVALID, "ABI version mismatch");
    if (!desc->n_enc_layers || !desc->n_dec_layers || !desc->enc_layers || !desc->dec_layers || !desc->dataset ||
        !desc->prior_loc || !desc->prior_scale || !desc->latent_dim || !desc->batch_size || !desc->n_features)
        return bsvi_fail(BSVI_ERR_INVALID, "incomplete amortised-model description");
    if (desc->batch_size > desc->dataset_size) return bsvi_fail(BSVI_ERR_INVALID, "batch_size exceeds dataset_size");
    if (desc->likelihood > BSVI_AMORT_LIK_NORMAL) return bsvi_fail(BSVI_ERR_UNSUPPORTED, "unknown likelihood kind");
    if (desc->likelihood == BSVI_AMORT_LIK_NORMAL) {
        if (!desc->likelihood_scale) return bsvi_fail(BSVI_ERR_INVALID, "a Normal likelihood needs its scale");
        for (uint32_t j = 0; j < desc->n_features; ++j)
            if (!(desc->likelihood_scale[j] > 0.0f)) return bsvi_fail(BSVI_ERR_INVALID, "likelihood scale must be positive");
    }
    for (uint32_t off : {desc->prior_loc_off, desc->prior_scale_off})
        if (off != BSVI_AMORT_CONSTANT && (size_t)off + desc->latent_dim > desc->n_params)
            return bsvi_fail(BSVI_ERR_INVALID, "learnable prior exceeds the parameter buffer");
    if (desc->lik_scale_off != BSVI_AMORT_CONSTANT) {
        if (desc->likelihood != BSVI_AMORT_LIK_NORMAL) return bsvi_fail(BSVI_ERR_INVALID, "a learnable likelihood scale needs the Normal likelihood");
        if (desc->lik_scale_size != 1u && desc->lik_scale_size != desc->n_features)
            return bsvi_fail(BSVI_ERR_INVALID, "a learnable likelihood scale has one value, or one per feature");
        if ((size_t)desc->lik_scale_off + desc->lik_scale_size > desc->n_params)
            return bsvi_fail(BSVI_ERR_INVALID, "learnable likelihood scale exceeds the parameter buffer");
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return bsvi_fail(BSVI_ERR_NO_DEVICE, "no HIP device");
    auto* a = new bsvi_amort();
    a->d = *desc;
    int rc = build_net(a->enc, desc->enc_layers, desc->n_enc_layers, (int)desc->n_features, desc->n_params);
    if (!rc) rc = build_net(a->dec, desc->dec_layers, desc->n_dec_layers, (int)desc->latent_dim, desc->n_params);
    auto bad = [&](const char* msg) { delete a; return bsvi_fail(BSVI_ERR_INVALID, msg); };
    if (rc) { delete a; return rc; }
    const uint32_t Dz = desc->latent_dim;
    auto head_ok = [&](uint32_t v, uint32_t col) {
        return v != 0 && v < a->enc.width.size() && (int)(col + Dz) <= a->enc.width[v];
    };
    if (!head_ok(desc->enc_loc_value, desc->enc_loc_col) || !head_ok(desc->enc_scale_value, desc->enc_scale_col) ||
        (desc->enc_loc_value == desc->enc_scale_value &&
         desc->enc_loc_col < desc->enc_scale_col + Dz && desc->enc_scale_col < desc->enc_loc_col + Dz))
        return bad("encoder heads must be two disjoint column ranges of width latent_dim in layer outputs");
    if (desc->dec_logits_value == 0 || desc->dec_logits_value >= a->dec.width.size() ||
        a->dec.width[desc->dec_logits_value] != (int)desc->n_features)
        return bad("decoder output must have width n_features");
    // the heads and the logits must be leaves: their gradients come from the row kernels only
    for (const auto& l : a->enc.layers)
        if (l.in_value == desc->enc_loc_value || l.in_value == desc->enc_scale_value) return bad("encoder heads must not feed further layers");
    for (const auto& l : a->dec.layers)
        if (l.in_value == desc->dec_logits_value) return bad("decoder logits must not feed further layers");
    if (desc->dec_scale_value) {
        if (desc->likelihood != BSVI_AMORT_LIK_NORMAL || desc->lik_scale_off != BSVI_AMORT_CONSTANT)
            return bad("a decoder-head scale needs the Normal likelihood and no learnable constant scale");
        if (desc->dec_scale_value >= a->dec.width.size() || desc->dec_scale_value == desc->dec_logits_value ||
            a->dec.width[desc->dec_scale_value] != (int)desc->n_features || a->dec.producer[desc->dec_scale_value] < 0)
            return bad("the decoder's scale head must be a layer output of width n_features");
        for (const auto& l : a->dec.layers)
            if (l.in_value == desc->dec_scale_value) return bad("the decoder's scale head must not feed further layers");
    }
    // workspace layout per row: values and gradients of every non-input value, z / dz, eps, rowf, rowlq, idx
    size_t off = 0;
    int alias_grad = -1;
    auto lay = [&](Net& net, bool input_buffers) {
        const size_t nv = net.width.size();
        net.val_off.assign(nv, 0);
        net.grad_off.assign(nv, 0);
        for (size_t v = input_buffers ? 0 : 1; v < nv; ++v) {
            net.val_off[v] = off;  off += net.ld[v];
            if (alias_grad == (int)v) { net.grad_off[v] = net.val_off[v]; continue; }
            net.grad_off[v] = off; off += net.ld[v];
        }
    };
    lay(a->enc, false);
    alias_grad = (int)desc->dec_logits_value;    // the logits are overwritten by their own gradient
    lay(a->dec, true);                           // decoder value 0 = z, its gradient = dz
    off += Dz;                  // eps
    off += 4;                   // rowf, rowlq, idx, spare
    a->floats_per_row = off;
    const size_t ds_bytes = (size_t)desc->dataset_size * desc->n_features * sizeof(float);
    if (hipMalloc(&a->dataset_dev, ds_bytes) != hipSuccess || hipMalloc(&a->prior_dev, 2 * Dz * sizeof(float)) != hipSuccess) {
        bsvi_amort_destroy(a);
        return bsvi_fail(BSVI_ERR_HIP, "hipMalloc of the dataset failed");
    }
    (void)hipMemcpy(a->dataset_dev, desc->dataset, ds_bytes, hipMemcpyHostToDevice);
    {   // is every value of the dataset exactly a bf16 (binarised images, pixel counts)?  Then keep a bf16 copy for xgemm_nt_kernel
        const char* xe = getenv("BSVI_AMORT_XGEMM");
        const size_t n_values = (size_t)desc->dataset_size * desc->n_features;
        const uint32_t* bits = reinterpret_cast<const uint32_t*>(desc->dataset);
        bool exact = !(xe && xe[0] == '0') && desc->n_features >= 64;      // (narrow inputs are not GEMM-bound)
        for (size_t i = 0; exact && i < n_values; ++i) exact = (bits[i] & 0xFFFFu) == 0u;
        if (exact) {
            const int Kp = ((int)desc->n_features + XBK - 1) / XBK * XBK;
            std::vector<uint16_t> host((size_t)desc->dataset_size * Kp, (uint16_t)0);
            for (size_t r = 0; r < desc->dataset_size; ++r)
                for (size_t k = 0; k < desc->n_features; ++k) host[r * Kp + k] = (uint16_t)(bits[r * desc->n_features + k] >> 16);
            bool ok = hipMalloc(&a->dataset_bf16_dev, host.size() * sizeof(uint16_t)) == hipSuccess;
            if (ok) ok = hipMemcpy(a->dataset_bf16_dev, host.data(), host.size() * sizeof(uint16_t), hipMemcpyHostToDevice) == hipSuccess;
            a->weight_pieces.assign(a->enc.layers.size(), nullptr);
            for (size_t l = 0; ok && l < a->enc.layers.size(); ++l)
                if (a->enc.layers[l].in_value == 0 && a->enc.layers[l].n_out > SKINNY)
                    ok = hipMalloc(&a->weight_pieces[l], 3 * (size_t)a->enc.layers[l].n_out * Kp * sizeof(uint16_t)) == hipSuccess;
            if (!ok) {
                bsvi_amort_destroy(a);
                return bsvi_fail(BSVI_ERR_HIP, "hipMalloc of the bf16 dataset copy failed");
            }
            a->data_exact = true;
            a->data_kp = Kp;
            const char* xd = getenv("BSVI_AMORT_XDW");
            a->xdw = !(xd && xd[0] == '0');
        }
    }
    if (desc->likelihood == BSVI_AMORT_LIK_NORMAL) {
        if (hipMalloc(&a->lik_scale_dev, desc->n_features * sizeof(float)) != hipSuccess ||
            hipMemcpy(a->lik_scale_dev, desc->likelihood_scale, desc->n_features * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            bsvi_amort_destroy(a);
            return bsvi_fail(BSVI_ERR_HIP, "hipMalloc of the likelihood scale failed");
        }
    }
    (void)hipMemcpy(a->prior_dev, desc->prior_loc, Dz * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemcpy(a->prior_dev + Dz, desc->prior_scale, Dz * sizeof(float), hipMemcpyHostToDevice);
    {   // wide layers whose input is a network value: six products of exact pieces on the bf16 matrix cores (x6gemm_kernel)
        const char* x6e = getenv("BSVI_AMORT_X6");
        const bool want = !(x6e && x6e[0] == '0');
        const Net* nets[2] = {&a->enc, &a->dec};
        for (int t = 0; t < 2 && want; ++t) {
            a->x6[t].assign(nets[t]->layers.size(), bsvi_amort::X6Layer{});
            for (size_t l = 0; l < nets[t]->layers.size(); ++l) {
                const auto& L = nets[t]->layers[l];
                const bool data_layer = t == 0 && L.in_value == 0;            // (gathered rows: the exact-data kernels or the f32 one)
                if (data_layer || L.n_in < 64 || L.n_out < 64 || (L.n_in & 3) || (L.n_out & 3)) continue;
                auto& X = a->x6[t][l];
                X.kp_nt = ((int)L.n_in + XBK - 1) / XBK * XBK;
                X.kp_nn = ((int)L.n_out + XBK - 1) / XBK * XBK;
                const bool ok = hipMalloc(&X.nt, 3 * (size_t)L.n_out * X.kp_nt * sizeof(uint16_t)) == hipSuccess &&
                                hipMalloc(&X.nn, 3 * (size_t)L.n_in * X.kp_nn * sizeof(uint16_t)) == hipSuccess;
                if (!ok) {
                    bsvi_amort_destroy(a);
                    return bsvi_fail(BSVI_ERR_HIP, "hipMalloc of the weight pieces failed");
                }
                a->x6_any = true;
            }
        }
    }
    const char* ov = getenv("BSVI_AMORT_OVERLAP");
    a->overlap = !(ov && ov[0] == '0');
    if (a->overlap) {
        bool ok = hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&a->joined, kEventFlags) == hipSuccess;
        a->ready.resize(desc->n_enc_layers + desc->n_dec_layers + 1, nullptr);      // (the last one: the transposed minibatch rows are written)
        for (auto& e : a->ready) ok = ok && hipEventCreateWithFlags(&e, kEventFlags) == hipSuccess;
        if (!ok) {
            bsvi_amort_destroy(a);
            return bsvi_fail(BSVI_ERR_HIP, "side stream / event creation failed");
        }
    }
    {   // do the layers' weights and biases tile [0, n_params) exactly?  Then nothing has to clear the gradient block
        std::vector<std::pair<size_t, size_t>> spans;
        for (const Net* net : {&a->enc, &a->dec})
            for (const auto& l : net->layers) {
                spans.emplace_back((size_t)l.weight_off, (size_t)l.n_in * l.n_out);
                if (l.bias_off != 0xFFFFFFFFu) spans.emplace_back((size_t)l.bias_off, (size_t)l.n_out);
            }
        for (uint32_t off : {desc->prior_loc_off, desc->prior_scale_off})
            if (off != BSVI_AMORT_CONSTANT) spans.emplace_back((size_t)off, (size_t)Dz);
        if (desc->lik_scale_off != BSVI_AMORT_CONSTANT) spans.emplace_back((size_t)desc->lik_scale_off, (size_t)desc->lik_scale_size);
        std::sort(spans.begin(), spans.end());
        size_t at = 0;
        bool exact = true;
        for (const auto& sp : spans) { exact = exact && sp.first == at; at = sp.first + sp.second; }
        a->layers_cover_params = exact && at == (size_t)desc->n_params;
        // the decoder's parameters (and the likelihood's scale, whose gradient is complete with them) as ONE range nothing else lies in
        size_t lo = SIZE_MAX, hi = 0;
        auto take = [&](size_t off, size_t n) { lo = std::min(lo, off); hi = std::max(hi, off + n); };
        for (const auto& l : a->dec.layers) {
            take(l.weight_off, (size_t)l.n_in * l.n_out);
            if (l.bias_off != 0xFFFFFFFFu) take(l.bias_off, l.n_out);
        }
        if (desc->lik_scale_off != BSVI_AMORT_CONSTANT) take(desc->lik_scale_off, desc->lik_scale_size);
        bool alone = lo < hi;
        auto outside = [&](size_t off, size_t n) { if (off < hi && off + n > lo) alone = false; };
        for (const auto& l : a->enc.layers) {
            outside(l.weight_off, (size_t)l.n_in * l.n_out);
            if (l.bias_off != 0xFFFFFFFFu) outside(l.bias_off, l.n_out);
        }
        for (uint32_t off : {desc->prior_loc_off, desc->prior_scale_off})
            if (off != BSVI_AMORT_CONSTANT) outside(off, Dz);
        if (alone) { a->bucket_first = (uint32_t)lo; a->bucket_count = (uint32_t)(hi - lo); }
    }
    a->d.enc_layers = a->enc.layers.data();
    a->d.dec_layers = a->dec.layers.data();
    a->d.dataset = nullptr;
    a->d.prior_loc = a->d.prior_scale = nullptr;
    a->d.likelihood_scale = nullptr;
    *out = a;
    return BSVI_OK;
}

// C[M][N] = X[rows[m]] W^T through xgemm_nt_kernel for the other kernel families of the library (the dense path's two
// products, dense_kernel.inc): X exact in bf16 [..][Kp], Wp the three bf16 pieces [3][N][Kp] of the other operand
int bsvi_xgemm_nt(const uint16_t* X, const int32_t* rows, const uint16_t* Wp, long plane_stride, float* C, int ldc,
                  int M, int N, int Kp, void* stream) {
    if (Kp % bsvi_amort_impl::XBK != 0 || M <= 0 || N <= 0) return bsvi_fail(BSVI_ERR_INVALID, "bsvi_xgemm_nt: bad shape");
    bsvi_amort_impl::XGemmArgs G{};
    G.X = X; G.rows = rows; G.Wp = Wp; G.plane_stride = plane_stride; G.C = C; G.ldc = ldc; G.M = M; G.N = N; G.Kp = Kp;
    bsvi_amort_impl::launch_xgemm(G, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, "xgemm_nt_kernel launch failed");
    return BSVI_OK;
}

// the same product with the result TRANSPOSED: Ct[n][m] (leading dimension ldct >= M), the split-k form of the kernel with one split
int bsvi_xgemm_nt_t(const uint16_t* X, const int32_t* rows, const uint16_t* Wp, long plane_stride, float* Ct, int ldct,
                    int M, int N, int Kp, void* stream) {
    if (Kp % bsvi_amort_impl::XBK != 0 || M <= 0 || N <= 0 || ldct < M) return bsvi_fail(BSVI_ERR_INVALID, "bsvi_xgemm_nt_t: bad shape");
    bsvi_amort_impl::XGemmArgs G{};
    G.X = X; G.rows = rows; G.Wp = Wp; G.plane_stride = plane_stride; G.C = Ct; G.ldc = ldct; G.M = M; G.N = N; G.Kp = Kp;
    G.steps_per_split = Kp / bsvi_amort_impl::XBK; G.part_stride = 0;
    G.rows_fastest = 3L * N > (long)M ? 1 : 0;
    const unsigned tiles = (unsigned)(((M + 127) / 128) * ((N + 127) / 128));
    hipLaunchKernelGGL((bsvi_amort_impl::xgemm_nt_glds_kernel<128, true>), dim3(tiles), dim3(256), 0, (hipStream_t)stream, G);
    if (hipGetLastError() != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, "xgemm_nt_glds_kernel launch failed");
    return BSVI_OK;
}

extern "C" int bsvi_amort_exact_data(const bsvi_amort* a) { return (a && a->data_exact) ? 1 : 0; }

extern "C" void bsvi_amort_destroy(bsvi_amort* a) {
    if (!a) return;
    if (a->dataset_dev) (void)hipFree(a->dataset_dev);
    if (a->dataset_bf16_dev) (void)hipFree(a->dataset_bf16_dev);
    for (auto p : a->weight_pieces)
        if (p) (void)hipFree(p);
    if (a->prior_dev) (void)hipFree(a->prior_dev);
    if (a->lik_scale_dev) (void)hipFree(a->lik_scale_dev);
    for (auto& v : a->x6)
        for (auto& l : v) {
            if (l.nt) (void)hipFree(l.nt);
            if (l.nn) (void)hipFree(l.nn);
        }
    for (auto e : a->ready)
        if (e) (void)hipEventDestroy(e);
    if (a->joined) (void)hipEventDestroy(a->joined);
    if (a->side) (void)hipStreamDestroy(a->side);
    for (auto e : a->bucket_ev)
        if (e) (void)hipEventDestroy(e);
    delete a;
}

// How a weight-gradient product C[M][N] = A[K][M]^T B[K][N] (K = all rows) is cut along K: every slice writes its own
// partial [M][N] (and partial column sums of A, [M]) and reduce_partials adds the slices in order.
struct TnPlan {
    bool skinny;
    int tiles, splits, chunk;      // MFMA path: output tiles, grid splits (whole splits per XCD; padding ones exit), rows per split
    int slices;                    // partials actually written
};
static TnPlan tn_plan(int M, int N, int K) {
    TnPlan p{};
    if (M <= SKINNY || N <= SKINNY) {
        p.skinny = true;
        p.chunk = 64;
        p.slices = (K + p.chunk - 1) / p.chunk;
        return p;
    }
    // about three workgroups per CU in flight, 64-row output tiles (less padding, half the splits of 128-row tiles);
    // at most 768 workgroups = three per CU in ONE round (a 769th would make some CU run four)
    p.tiles = ((M + 63) / 64) * ((N + BN - 1) / BN);
    int splits = std::max(1, std::min((K + BK - 1) / BK, 768 / std::max(p.tiles, 1)));
    p.chunk = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    p.slices = (K + p.chunk - 1) / p.chunk;
    p.splits = (p.slices + 7) / 8 * 8;
    return p;
}
static size_t align4(size_t n) { return (n + 3) / 4 * 4; }      // partial regions start on 16-byte boundaries
static size_t tn_partial_floats(int M, int N, int K, bool bias) {
    const TnPlan p = tn_plan(M, N, K);
    // (the six-piece form of the same product, x6tn_kernel, cuts the rows into its own number of slices: room for either)
    const size_t slices = p.skinny ? (size_t)p.slices : std::max<size_t>((size_t)p.slices, (size_t)x6tn_plan(M, N, K).slices);
    return align4(slices * M * N) + (bias ? align4(slices * M) : 0);
}

// floats of the partial sums behind the per-row part of the workspace: one slice set per Linear layer (weights + bias)
// and the workgroup partials of the loss sums
// does encoder layer li take its weight gradient through the exact-data product at R rows?
static bool xdw_layer(const bsvi_amort* a, size_t li, size_t R) {
    if (!a->xdw || li >= a->weight_pieces.size() || !a->weight_pieces[li]) return false;
    const size_t Rp = (R + 63) / 64 * 64;
    const auto& l = a->enc.layers[li];
    return std::max<size_t>(3 * (size_t)l.n_out, l.n_in) * Rp * 2 < 0xF0000000ull;      // 32-bit byte offsets in the kernel
}
static size_t xdw_floats(const bsvi_amort* a, size_t li, size_t R) {      // slice partials, bias partials, the pieces of dY
    const auto& l = a->enc.layers[li];
    const XdwPlan p = xdw_plan((int)l.n_in, (int)l.n_out, R);
    return align4((size_t)p.slices * l.n_out * l.n_in) + align4((size_t)(p.Rp / 64) * l.n_out) + align4(3 * (size_t)l.n_out * p.Rp / 2);
}
// row slices of amort_lik_scale_grad: enough workgroups to fill the chip, at least 64 rows each
static uint32_t lik_scale_slices(size_t R) {
    const size_t s = (R + 63) / 64;
    return (uint32_t)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

static size_t partial_floats(const bsvi_amort* a, size_t R) {
    size_t n = align4(2 * ((R + 255) / 256));
    bool any_xdw = false;
    for (const Net* net : {&a->enc, &a->dec})
        for (size_t li = 0; li < net->layers.size(); ++li) {
            const auto& l = net->layers[li];
            if (net == &a->enc && xdw_layer(a, li, R)) { n += xdw_floats(a, li, R); any_xdw = true; }
            else n += tn_partial_floats((int)l.n_out, (int)l.n_in, (int)R, l.bias_off != 0xFFFFFFFFu);
        }
    if (any_xdw) n += align4((size_t)a->d.n_features * ((R + 63) / 64 * 64) / 2);      // the transposed minibatch rows (bf16)
    if (a->d.prior_loc_off != BSVI_AMORT_CONSTANT || a->d.prior_scale_off != BSVI_AMORT_CONSTANT)
        n += 2 * align4(((R + 255) / 256) * (size_t)a->d.latent_dim);                  // a learnable prior's gradient partials
    if (a->d.lik_scale_off != BSVI_AMORT_CONSTANT)
        n += align4((size_t)lik_scale_slices(R) * a->d.n_features);                    // a learnable likelihood scale's
    n += align4((size_t)((a->d.n_features + 127) / 128) * R);                           // the fused likelihood epilogue's partial log-likelihoods [column tiles][R]
    return n + 16;
}

extern "C" size_t bsvi_amort_workspace_bytes(const bsvi_amort* a, uint32_t n_samples_local) {
    if (!a) return 0;
    const size_t R = (size_t)n_samples_local * a->d.batch_size;
    return (R * a->floats_per_row + 64 + partial_floats(a, R)) * sizeof(float);
}

// steps of global loads kept in flight by the MFMA kernels, per operand layout (BSVI_GEMM_PF=<nt><nn><tn>, e.g. 112)
static int prefetch_depth(int mode) {
    static const std::string cfg = [] { const char* e = getenv("BSVI_GEMM_PF"); return std::string(e && strlen(e) == 3 ? e : "222"); }();
    return cfg[mode] == '2' ? 2 : 1;
}

// rows from which the 128-row tiles of x6gemm_kernel fill the chip better than the f32 kernel's 64-row ones
constexpr int kX6MinRows = 256;
static int launch_x6(bool nn, const X6Args& X, hipStream_t stream) {
    const int tiles = ((X.M + 127) / 128) * ((X.N + 127) / 128);
    static const int dbg = [] { const char* e = getenv("BSVI_X6_DEBUG"); return e ? atoi(e) : 0; }();
    // round 6's kernel (LDS-DMA staging, 16-byte stores) wants 16-byte aligned rows of every matrix it touches with wide accesses;
    // anything else — and BSVI_X6_V=5 — takes round 5's kernel, which stages through registers
    static const bool v6 = [] { const char* e = getenv("BSVI_X6_V"); return !(e && e[0] == '5'); }();
    auto al16 = [](const void* p, int ld) { return ((uintptr_t)p & 15) == 0 && (ld & 3) == 0; };
    const bool wide_ok = v6 && (X.N & 3) == 0 && al16(X.A, X.lda) && al16(X.C, X.ldc) && (!X.Y || al16(X.Y, X.ldy)) && ((uintptr_t)X.Bp & 15) == 0;
    static const int var = [] { const char* e = getenv("BSVI_X6_VAR"); return e ? atoi(e) : 1; }();      // 1: the hand-ordered main loop (default), 0: the compiler's order
    if (wide_ok && ((size_t)(X.M - 1) * X.lda + X.K) * sizeof(float) < 0x7fffffffull && 3 * (size_t)X.plane_stride * 2 < 0x7fffffffull) {
#define BSVI_X6_LAUNCH(NNV, D, V) hipLaunchKernelGGL((x6gemm_kernel<NNV, D, V>), dim3(tiles), dim3(256), 0, stream, X)
        if (!nn && dbg == 2) BSVI_X6_LAUNCH(false, 2, 0);
        else if (!nn && dbg == 3) BSVI_X6_LAUNCH(false, 3, 0);
        else if (!nn && dbg == 4) BSVI_X6_LAUNCH(false, 4, 0);
        else if (!nn && dbg == 5) BSVI_X6_LAUNCH(false, 5, 0);
        else if (!nn && dbg == 9) BSVI_X6_LAUNCH(false, 9, 1);
        else if (nn && dbg == 12) BSVI_X6_LAUNCH(true, 2, 0);
        else if (nn && dbg == 13) BSVI_X6_LAUNCH(true, 3, 0);
        else if (!nn && X.lik_x) hipLaunchKernelGGL((x6gemm_kernel<false, 0, 1, true>), dim3(tiles), dim3(256), 0, stream, X);
        else if (var == 1) { if (nn) BSVI_X6_LAUNCH(true, 0, 1); else BSVI_X6_LAUNCH(false, 0, 1); }
        else if (nn) BSVI_X6_LAUNCH(true, 0, 0);
        else BSVI_X6_LAUNCH(false, 0, 0);
#undef BSVI_X6_LAUNCH
    } else if (nn) hipLaunchKernelGGL((x6gemm_r5_kernel<true>), dim3(tiles), dim3(256), 0, stream, X);
    else hipLaunchKernelGGL((x6gemm_r5_kernel<false>), dim3(tiles), dim3(256), 0, stream, X);
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}

static int launch_gemm(int mode, GemmArgs G, hipStream_t stream) {
    if (G.M <= 0 || G.N <= 0 || G.K <= 0) return BSVI_OK;
    auto aligned = [](const void* p, int ld) { return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0); };
    G.vecA = aligned(G.A, G.lda);
    G.vecB = aligned(G.B, G.ldb);
    // a side of width <= 8: the memory-bound kernels
    static const bool second_generation = [] { const char* e = getenv("BSVI_NARROW_GEN"); return !(e && e[0] == '1'); }();
    if (mode == MODE_TN && (G.M <= SKINNY || G.N <= SKINNY)) {
        const int narrow_is_a = G.M <= G.N ? 1 : 0;
        const int wide = narrow_is_a ? G.N : G.M, narrow = narrow_is_a ? G.M : G.N;
        const int rows_per_block = tn_plan(G.M, G.N, G.K).chunk;
        const bool wide_vec = narrow_is_a ? G.vecB : G.vecA, narrow_vec = narrow_is_a ? G.vecA : G.vecB;
        const bool out_vec = !narrow_is_a || (G.ldc % 4 == 0 && (uintptr_t)G.C % 16 == 0 && G.part_stride % 4 == 0);
        const bool bias_vec = !G.bias_grad || narrow_is_a || (G.M % 4 == 0 && (uintptr_t)G.bias_grad % 16 == 0);
        if (second_generation && wide % 4 == 0 && wide_vec && narrow_vec && out_vec && bias_vec) {
            // (the padded leading dimension of the narrow operand covers the 16-byte loads: ld >= 4 ceil(narrow / 4))
            dim3 grid((wide / 4 + 255) / 256, (G.K + rows_per_block - 1) / rows_per_block);
            if (narrow <= 4) hipLaunchKernelGGL((outer_kernel<4>), grid, dim3(256), 0, stream, G, narrow_is_a, rows_per_block);
            else hipLaunchKernelGGL((outer_kernel<8>), grid, dim3(256), 0, stream, G, narrow_is_a, rows_per_block);
            HIP_TRY(hipGetLastError());
            return BSVI_OK;
        }
        dim3 grid((wide + 255) / 256, (G.K + rows_per_block - 1) / rows_per_block);
        hipLaunchKernelGGL(skinny_tn_kernel, grid, dim3(256), 0, stream, G, narrow_is_a, rows_per_block);
        HIP_TRY(hipGetLastError());
        return BSVI_OK;
    }
    if (mode != MODE_TN) {
        const int sbk = mode == MODE_NT ? 1 : G.ldb, sbn = mode == MODE_NT ? G.ldb : 1;
        if (second_generation && G.N <= SKINNY && G.K % 4 == 0 && G.vecA && (size_t)G.K * SKINNY * sizeof(float) <= 64 * 1024) {
            const int rows_per_block = 32;
            const dim3 grid((G.M + rows_per_block - 1) / rows_per_block);
            if (G.N <= 2) hipLaunchKernelGGL((rowdot_kernel<2>), grid, dim3(256), (size_t)2 * G.K * sizeof(float), stream, G, sbk, sbn, mode, rows_per_block);
            else if (G.N <= 4) hipLaunchKernelGGL((rowdot_kernel<4>), grid, dim3(256), (size_t)4 * G.K * sizeof(float), stream, G, sbk, sbn, mode, rows_per_block);
            else hipLaunchKernelGGL((rowdot_kernel<8>), grid, dim3(256), (size_t)8 * G.K * sizeof(float), stream, G, sbk, sbn, mode, rows_per_block);
            HIP_TRY(hipGetLastError());
            return BSVI_OK;
        }
        if (G.N <= SKINNY && (size_t)G.N * G.K <= 8192) {
            static const int rows_per_block = [] { const char* e = getenv("BSVI_SKINNY_ROWS"); return e ? atoi(e) : 16; }();
            hipLaunchKernelGGL(skinny_n_kernel, dim3((G.M + rows_per_block - 1) / rows_per_block), dim3(256),
                               (size_t)G.N * G.K * sizeof(float), stream, G, sbk, sbn, mode, rows_per_block);
            HIP_TRY(hipGetLastError());
            return BSVI_OK;
        }
        if (G.K <= SKINNY && (uint64_t)G.M * (uint64_t)G.N < (1ull << 32) - 256) {
            auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
            if (second_generation && mode == MODE_NT && G.ldb == G.K && (G.N & 3) == 0 && al16(G.B) && (G.ldc & 3) == 0 && al16(G.C) &&
                (!G.bias || al16(G.bias))) {
                const dim3 grid((unsigned)(((long)G.M * (G.N >> 2) + 255) / 256));
                switch (G.K) {
#define BSVI_K4NT(KK) case KK: hipLaunchKernelGGL((skinny_k4nt_kernel<KK>), grid, dim3(256), 0, stream, G); break;
                    BSVI_K4NT(1) BSVI_K4NT(2) BSVI_K4NT(3) BSVI_K4NT(4) BSVI_K4NT(5) BSVI_K4NT(6) BSVI_K4NT(7) BSVI_K4NT(8)
#undef BSVI_K4NT
                }
                HIP_TRY(hipGetLastError());
                return BSVI_OK;
            }
            // (B stored [N][K], the forward layout, would need strided scalar loads per output: measured slower)
            const bool vec4 = sbn == 1 && (sbk & 3) == 0 && al16(G.B) && (G.N & 3) == 0 && (G.ldc & 3) == 0 && al16(G.C) &&
                              (!G.bias || al16(G.bias)) && (!G.Y || ((G.ldy & 3) == 0 && al16(G.Y)));
            if (vec4) {
                const long quads = (long)G.M * (G.N >> 2);
                hipLaunchKernelGGL(skinny_k4_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, G, sbk, sbn, mode);
                HIP_TRY(hipGetLastError());
                return BSVI_OK;
            }
            const long total = (long)G.M * G.N;
            hipLaunchKernelGGL(skinny_k_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, G, sbk, sbn, mode);
            HIP_TRY(hipGetLastError());
            return BSVI_OK;
        }
    }
    const int tiles_n = (G.N + BN - 1) / BN;
    int tiles = ((G.M + BM - 1) / BM) * tiles_n;
    if (mode == MODE_TN) {
        const TnPlan plan = tn_plan(G.M, G.N, G.K);
        tiles = plan.tiles;
        const int splits = plan.splits;
        G.k_chunk = plan.chunk;
        G.tiles = tiles;
        G.remap = 1;
        dim3 grid(tiles * splits, 1, 1);
        if (prefetch_depth(mode) == 2) hipLaunchKernelGGL((gemm_kernel<MODE_TN, 64, 2>), grid, dim3(NTHREADS), 0, stream, G);
        else hipLaunchKernelGGL((gemm_kernel<MODE_TN, 64, 1>), grid, dim3(NTHREADS), 0, stream, G);
    } else {
        // (round 4, measured and removed: stream-K — a grid of the resident slots, equal runs of (128 x 128 tile, k step) units
        //  per workgroup, a cut tile finished by the later workgroup in a fixed order.  cfg 5's layers at 25 600 rows: 136 /
        //  117 / 94 / 179 us forward and 108 / 158 / 140 us input gradient against 126 / 81 / 84 / 125 and 89 / 93 / 118 here.
        //  With K = 256 ... 784 a tile is 16 ... 49 steps: what these shapes pay for is the tile's prologue and its 64 scalar
        //  stores per thread, which four co-resident 64-row workgroups hide from each other and three persistent 128-row ones
        //  do not — not the uneven last round the shape count suggested.)
        // 64-row tiles when 128-row tiles would leave most CUs with one or two workgroups
        static const int half_below = [] { const char* e = getenv("BSVI_GEMM_HALF_BELOW"); return e ? atoi(e) : 1536; }();
        const bool half = tiles < half_below && G.M > 64;
        if (half) tiles = ((G.M + 63) / 64) * tiles_n;
        G.remap = (tiles % 8 == 0) ? 1 : 0;
        dim3 grid(tiles, 1, 1);
        const bool deep = prefetch_depth(mode) == 2;
        if (mode == MODE_NT) {
            if (half && deep) hipLaunchKernelGGL((gemm_kernel<MODE_NT, 64, 2>), grid, dim3(NTHREADS), 0, stream, G);
            else if (half) hipLaunchKernelGGL((gemm_kernel<MODE_NT, 64, 1>), grid, dim3(NTHREADS), 0, stream, G);
            else if (deep) hipLaunchKernelGGL((gemm_kernel<MODE_NT, 128, 2>), grid, dim3(NTHREADS), 0, stream, G);
            else hipLaunchKernelGGL((gemm_kernel<MODE_NT, 128, 1>), grid, dim3(NTHREADS), 0, stream, G);
        } else {
            if (half && deep) hipLaunchKernelGGL((gemm_kernel<MODE_NN, 64, 2>), grid, dim3(NTHREADS), 0, stream, G);
            else if (half) hipLaunchKernelGGL((gemm_kernel<MODE_NN, 64, 1>), grid, dim3(NTHREADS), 0, stream, G);
            else if (deep) hipLaunchKernelGGL((gemm_kernel<MODE_NN, 128, 2>), grid, dim3(NTHREADS), 0, stream, G);
            else hipLaunchKernelGGL((gemm_kernel<MODE_NN, 128, 1>), grid, dim3(NTHREADS), 0, stream, G);
        }
    }
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}

// the f32-input MFMA products for the other kernel families of the library (the Bayesian-neural-network path on data that is not
// exactly bf16, bnn_kernel.inc): no bias, no activation
int bsvi_gemm_f32(int mode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, void* stream) {
    if ((mode != MODE_NT && mode != MODE_NN) || !A || !B || !C) return bsvi_fail(BSVI_ERR_INVALID, "bsvi_gemm_f32: bad arguments");
    GemmArgs G{};
    G.A = A; G.B = B; G.C = C; G.M = M; G.N = N; G.K = K; G.lda = lda; G.ldb = ldb; G.ldc = ldc;
    return launch_gemm(mode, G, (hipStream_t)stream);
}

extern "C" int bsvi_debug_gemm(int mode, const float* a_dev, const float* b_dev, float* c_dev, const int32_t* rows_dev,
                               uint32_t m, uint32_t n, uint32_t k, uint32_t lda, uint32_t ldb, uint32_t ldc,
                               const float* bias_or_y_dev, uint32_t ldy, uint32_t activation, float post_add,
                               uint32_t accumulate, void* stream) {
    if (mode == 3) {
        // the exact-data forward product C = act(A[rows] B^T + bias) through xgemm_nt_kernel: A (whose values must be exact
        // in bf16 — not checked here) is converted, B split into its three pieces, in buffers the hook keeps
        if (!a_dev || !b_dev || !c_dev) return bsvi_fail(BSVI_ERR_INVALID, "bad gemm arguments");
        const int Kp = ((int)k + XBK - 1) / XBK * XBK;
        uint32_t a_rows = m;                       // with a gather the caller passes the number of SOURCE rows in `accumulate`
        if (rows_dev) a_rows = accumulate;
        static uint16_t* xb = nullptr; static size_t xb_n = 0;
        static uint16_t* wp = nullptr; static size_t wp_n = 0;
        const size_t need_x = (size_t)a_rows * Kp, need_w = 3 * (size_t)n * Kp;
        if (need_x > xb_n) { (void)hipDeviceSynchronize(); if (xb) (void)hipFree(xb); xb = nullptr; HIP_TRY(hipMalloc(&xb, need_x * 2)); xb_n = need_x; }
        if (need_w > wp_n) { (void)hipDeviceSynchronize(); if (wp) (void)hipFree(wp); wp = nullptr; HIP_TRY(hipMalloc(&wp, need_w * 2)); wp_n = need_w; }
        hipStream_t st = (hipStream_t)stream;
        if (lda != k || ldb != k) return bsvi_fail(BSVI_ERR_INVALID, "mode 3 takes densely stored operands");
        hipLaunchKernelGGL(x_to_bf16_kernel, dim3((unsigned)((need_x + 255) / 256)), dim3(256), 0, st, a_dev, (long)a_rows, (int)k, Kp, xb);
        hipLaunchKernelGGL(xw_split_kernel, dim3((unsigned)(((size_t)n * Kp + 255) / 256)), dim3(256), 0, st, b_dev, (int)n, (int)k, Kp, wp);
        XGemmArgs X{};
        X.X = xb; X.rows = rows_dev; X.Wp = wp; X.plane_stride = (long)n * Kp; X.C = c_dev; X.ldc = (int)ldc;
        X.M = (int)m; X.N = (int)n; X.Kp = Kp; X.bias = bias_or_y_dev; X.act = (int)activation; X.post_add = post_add;
        launch_xgemm(X, st);
        HIP_TRY(hipGetLastError());
        return BSVI_OK;
    }
    if (mode == 4) {
        // the exact-data weight gradient C[M][N] = A[K][M]^T B[rows[k]][N] (and the column sums of A into bias_or_y_dev):
        // B (exact in bf16, not checked; `accumulate` source rows, stored densely) is converted, then the product path's
        // three launches and the reduction over the slices
        if (!a_dev || !b_dev || !c_dev || !rows_dev) return bsvi_fail(BSVI_ERR_INVALID, "bad gemm arguments");
        if (ldb != n) return bsvi_fail(BSVI_ERR_INVALID, "mode 4 takes a densely stored B");
        const int Kp = ((int)n + XBK - 1) / XBK * XBK;
        const XdwPlan plan = xdw_plan((int)n, (int)m, k);
        float* bias_acc = const_cast<float*>(bias_or_y_dev);
        static uint16_t* xb = nullptr; static size_t xb_n = 0;
        static float* part = nullptr; static size_t part_n = 0;
        const size_t need_x = (size_t)accumulate * Kp;
        const size_t f_slices = align4((size_t)plan.slices * m * n), f_col = align4((size_t)(plan.Rp / 64) * m);
        const size_t f_t = align4(3 * (size_t)m * plan.Rp / 2), f_xt = align4((size_t)n * plan.Rp / 2);
        const size_t need_p = f_slices + f_col + f_t + f_xt;
        if (need_x > xb_n) { (void)hipDeviceSynchronize(); if (xb) (void)hipFree(xb); xb = nullptr; xb_n = 0; HIP_TRY(hipMalloc(&xb, need_x * 2)); xb_n = need_x; }
        if (need_p > part_n) { (void)hipDeviceSynchronize(); if (part) (void)hipFree(part); part = nullptr; part_n = 0; HIP_TRY(hipMalloc(&part, need_p * 4)); part_n = need_p; }
        hipStream_t st = (hipStream_t)stream;
        float* slices = part; float* colsum = part + f_slices;
        uint16_t* T = reinterpret_cast<uint16_t*>(colsum + f_col);
        uint16_t* XT = reinterpret_cast<uint16_t*>(colsum + f_col + f_t);
        hipLaunchKernelGGL(x_to_bf16_kernel, dim3((unsigned)((need_x + 255) / 256)), dim3(256), 0, st, b_dev, (long)accumulate, (int)n, Kp, xb);
        hipLaunchKernelGGL(xt_gather_kernel, dim3((unsigned)(plan.Rp / 64), (unsigned)((n + 127) / 128)), dim3(256), 0, st,
                           xb, rows_dev, Kp, (int)n, (int)k, plan.Rp, XT);
        hipLaunchKernelGGL(dy_split_t_kernel, dim3((unsigned)(plan.Rp / 64), (unsigned)((m + 63) / 64)), dim3(256), 0, st,
                           a_dev, (int)lda, (int)k, plan.Rp, (int)m, T, bias_acc ? colsum : nullptr);
        launch_xdw(XT, T, (int)n, (int)m, plan, slices, st);
        SegmentTable S{};
        S.seg[0] = Segment{c_dev, slices, m, n, ldc, (uint32_t)plan.slices, 0u, 0u, 0u, 0u};
        S.n = 1;
        uint32_t blocks = segment_blocks(S.seg[0]);
        if (bias_acc) {
            S.seg[1] = Segment{bias_acc, colsum, 1u, m, m, (uint32_t)(plan.Rp / 64), blocks, 0u, 0u, 0u};
            S.n = 2;
            blocks += segment_blocks(S.seg[1]);
        }
        hipLaunchKernelGGL(reduce_partials, dim3(blocks), dim3(256), 0, st, S);
        HIP_TRY(hipGetLastError());
        return BSVI_OK;
    }
    if (mode == 8) {      // diagnostics: the stamps of the last BSVI_X6_DEBUG=9 launch, m values of 8 bytes into c_dev
        if (!c_dev || m > 8u * 8192u) return bsvi_fail(BSVI_ERR_INVALID, "bad gemm arguments");
        HIP_TRY(hipMemcpyFromSymbol(c_dev, HIP_SYMBOL(x6_stamps), (size_t)m * 8, 0, hipMemcpyDeviceToDevice));
        return BSVI_OK;
    }
    if (mode == 5 || mode == 6) {
        // modes 0 / 1 through x6gemm_kernel: six products of exact bf16 pieces (B's pieces split here, in a buffer the hook keeps)
        if (!a_dev || !b_dev || !c_dev || rows_dev || (k & 3) || (lda & 3)) return bsvi_fail(BSVI_ERR_INVALID, "bad gemm arguments");
        const int Kp = ((int)k + XBK - 1) / XBK * XBK;
        static uint16_t* wp = nullptr; static size_t wp_n = 0;
        const size_t need_w = 3 * (size_t)n * Kp;
        if (need_w > wp_n) { (void)hipDeviceSynchronize(); if (wp) (void)hipFree(wp); wp = nullptr; wp_n = 0; HIP_TRY(hipMalloc(&wp, need_w * 2)); wp_n = need_w; }
        hipStream_t st = (hipStream_t)stream;
        if (mode == 5 ? ldb != k : ldb != n) return bsvi_fail(BSVI_ERR_INVALID, "modes 5 / 6 take a densely stored B");
        X6SplitTable T{};
        T.e[0] = {b_dev, wp, (int)n, (int)k, Kp, mode == 6 ? 1 : 0, 0u};
        T.n = 1;
        hipLaunchKernelGGL(x6_split_kernel, dim3((unsigned)(((size_t)n * Kp + 255) / 256)), dim3(256), 0, st, T);
        X6Args X{};
        X.A = a_dev; X.lda = (int)lda; X.Bp = wp; X.Kp = Kp; X.plane_stride = (long)n * Kp;
        X.C = c_dev; X.ldc = (int)ldc; X.M = (int)m; X.N = (int)n; X.K = (int)k;
        X.act = (int)activation; X.post_add = post_add; X.accumulate = (int)accumulate;
        if (mode == 5) X.bias = bias_or_y_dev; else { X.Y = bias_or_y_dev; X.ldy = (int)ldy; }
        return launch_x6(mode == 6, X, st);
    }
    if (mode == 7) {
        // mode 2 (C[m][n] = sum_k A[k][m] B[k][n], column sums of A into bias_or_y_dev) through x6tn_kernel: six products of exact pieces,
        // both operands split and transposed on the way into LDS
        // (with rows_dev: B's rows are gathered and B is taken as exactly bf16 — the data layer's form, one piece and three products)
        if (!a_dev || !b_dev || !c_dev) return bsvi_fail(BSVI_ERR_INVALID, "bad gemm arguments");
        float* bias_acc = const_cast<float*>(bias_or_y_dev);
        const X6TnPlan xp = x6tn_plan((int)m, (int)n, (int)k);
        static float* part = nullptr;
        static size_t part_floats = 0;
        const size_t need = align4((size_t)xp.slices * m * n) + align4((size_t)xp.slices * m);
        if (need > part_floats) {
            (void)hipDeviceSynchronize();
            if (part) (void)hipFree(part);
            part = nullptr; part_floats = 0;
            HIP_TRY(hipMalloc(&part, need * sizeof(float)));
            part_floats = need;
        }
        X6TnArgs X{};
        X.A = a_dev; X.lda = (int)lda; X.B = b_dev; X.ldb = (int)ldb; X.C = part; X.part_stride = (long)m * n;
        X.bias_grad = bias_acc ? part + align4((size_t)xp.slices * m * n) : nullptr;
        X.M = (int)m; X.N = (int)n; X.K = (int)k; X.chunk = xp.chunk; X.tiles = xp.tiles; X.rows = rows_dev;
        if (rows_dev) hipLaunchKernelGGL((x6tn_kernel<0, true>), dim3((unsigned)(xp.tiles * xp.slices)), dim3(256), 0, (hipStream_t)stream, X);
        else hipLaunchKernelGGL((x6tn_kernel<0>), dim3((unsigned)(xp.tiles * xp.slices)), dim3(256), 0, (hipStream_t)stream, X);
        SegmentTable T{};
        T.seg[0] = Segment{c_dev, part, m, n, ldc, (uint32_t)xp.slices, 0u, 0u, 0u, 0u};       // (C = the product: nothing kept)
        T.n = 1;
        uint32_t blocks = segment_blocks(T.seg[0]);
        if (bias_acc) {
            T.seg[1] = Segment{bias_acc, X.bias_grad, 1u, m, m, (uint32_t)xp.slices, blocks, 0u, 0u, 0u};
            T.n = 2;
            blocks += segment_blocks(T.seg[1]);
        }
        hipLaunchKernelGGL(reduce_partials, dim3(blocks), dim3(256), 0, (hipStream_t)stream, T);
        HIP_TRY(hipGetLastError());
        return BSVI_OK;
    }
    if (mode < 0 || mode > 2 || !a_dev || !b_dev || !c_dev) return bsvi_fail(BSVI_ERR_INVALID, "bad gemm arguments");
    GemmArgs G{};
    G.A = a_dev; G.B = b_dev; G.C = c_dev; G.rows = rows_dev;
    G.M = (int)m; G.N = (int)n; G.K = (int)k; G.lda = (int)lda; G.ldb = (int)ldb; G.ldc = (int)ldc;
    G.act = (int)activation; G.post_add = post_add; G.accumulate = (int)accumulate;
    if (mode == MODE_NT) G.bias = bias_or_y_dev;
    if (mode == MODE_NN) { G.Y = bias_or_y_dev; G.ldy = (int)ldy; }
    if (mode != MODE_TN) return launch_gemm(mode, G, (hipStream_t)stream);
    // the product path keeps the partials in its workspace; the hook borrows a buffer for them
    float* bias_acc = const_cast<float*>(bias_or_y_dev);   // [M] accumulator of the column sums of A, or NULL
    const TnPlan plan = tn_plan(G.M, G.N, G.K);
    static float* part = nullptr;          // grow-only, kept for the life of the process (a test / probe hook)
    static size_t part_floats = 0;
    const size_t need = tn_partial_floats(G.M, G.N, G.K, bias_acc != nullptr);
    if (need > part_floats) {
        (void)hipDeviceSynchronize();
        if (part) (void)hipFree(part);
        part = nullptr; part_floats = 0;
        HIP_TRY(hipMalloc(&part, need * sizeof(float)));
        part_floats = need;
    }
    G.C = part; G.ldc = G.N; G.part_stride = (long)G.M * G.N;
    G.bias_grad = bias_acc ? part + align4((size_t)plan.slices * G.M * G.N) : nullptr;
    int rc = launch_gemm(mode, G, (hipStream_t)stream);
    if (!rc) {
        SegmentTable T{};
        T.seg[0] = Segment{c_dev, part, m, n, ldc, (uint32_t)plan.slices, 0u, 1u, 0u, 0u};
        T.n = 1;
        uint32_t blocks = segment_blocks(T.seg[0]);
        if (bias_acc) {
            T.seg[1] = Segment{bias_acc, G.bias_grad, 1u, m, m, (uint32_t)plan.slices, blocks, 1u, 0u, 0u};
            T.n = 2;
            blocks += segment_blocks(T.seg[1]);
        }
        hipLaunchKernelGGL(reduce_partials, dim3(blocks), dim3(256), 0, (hipStream_t)stream, T);
        if (hipGetLastError() != hipSuccess) rc = bsvi_fail(BSVI_ERR_HIP, "reduce_partials launch failed");
    }
    return rc;
}

extern "C" int bsvi_amort_fwd_bwd(const bsvi_amort* a, const bsvi_amort_args* args) {
    if (args) BSVI_CHECK_STRUCT(args, bsvi_amort_args);
    if (!a || !args || !args->params_dev || !args->out_dev || !args->workspace_dev || !args->n_samples_local)
        return bsvi_fail(BSVI_ERR_INVALID, "null argument");
    if (args->estimator > 1) return bsvi_fail(BSVI_ERR_UNSUPPORTED, "estimator must be 0 (pathwise) or 1 (BlackBox)");
    hipStream_t stream = (hipStream_t)args->stream;
    const bsvi_amort_desc& d = a->d;
    const int B = (int)d.batch_size, Dz = (int)d.latent_dim, P = (int)d.n_features;
    const size_t R = (size_t)args->n_samples_local * B;
    if (R > 0x7fffffffu / 1024) return bsvi_fail(BSVI_ERR_RESOURCE, "too many rows for one launch");
    float* ws = (float*)args->workspace_dev;
    float* out = args->out_dev;
    const float* params = args->params_dev;
    float* grads = out + BSVI_OUT_HEADER;

    auto val = [&](const Net& net, uint32_t v) { return ws + net.val_off[v] * R; };
    auto grad = [&](const Net& net, uint32_t v) { return ws + net.grad_off[v] * R; };
    size_t tail = 0;
    for (const Net* net : {&a->enc, &a->dec})
        for (size_t v = 0; v < net->width.size(); ++v) tail = std::max(tail, std::max(net->val_off[v], net->grad_off[v]) + net->ld[v]);
    float* eps = ws + tail * R;
    float* rowf = eps + (size_t)Dz * R;
    float* rowlq = rowf + R;
    int32_t* idx = (int32_t*)(rowlq + R);

    // (out[0], out[1] and every gradient element are WRITTEN by reduce_partials, out[2], out[3] by the finalize step)
    if (!a->layers_cover_params) HIP_TRY(hipMemsetAsync(out, 0, (BSVI_OUT_HEADER + (size_t)d.n_params) * sizeof(float), stream));

    // partial sums (second part of the workspace) and the table reduce_partials works through at the end
    float* part = ws + (R * a->floats_per_row + 64 + 3) / 4 * 4;
    SegmentTable segments{};
    uint32_t reduce_blocks = 0;
    auto add_segment = [&](float* dst, const float* src, uint32_t rows, uint32_t cols, uint32_t splits) -> int {
        if (segments.n == kMaxSegments) return bsvi_fail(BSVI_ERR_RESOURCE, "too many layers for one reduction launch");
        Segment& S = segments.seg[segments.n++];
        S = Segment{dst, src, rows, cols, cols, splits, reduce_blocks, 0u, 0u, 0u};
        reduce_blocks += segment_blocks(S);
        return BSVI_OK;
    };

    RowParams D{};
    D.R = (int)R; D.B = B; D.DS = (int)d.dataset_size; D.P = P; D.Dz = Dz;
    D.n_local = (int)args->n_samples_local; D.sample_base = (int)args->sample_base;
    D.seed_lo = (uint32_t)args->seed; D.seed_hi = (uint32_t)(args->seed >> 32);
    D.off_lo = (uint32_t)args->offset; D.off_hi = (uint32_t)(args->offset >> 32);
    D.estimator = (int)args->estimator;
    // EmpiricalDistribution._get_entropy (distributions.py:464-473) is Categorical(ones(n)).entropy() with
    // n = dataset.shape[0]; the posterior's dataset has been tiled to number_samples rows, so every row's entropy
    // carries the constant log(number_samples).  Kept: it shifts the loss and weights BlackBox's score term.
    D.entropy_const = logf((float)args->n_samples_global);
    D.indices_in = args->indices_dev; D.idx = idx; D.indices_out = args->indices_out_dev;
    D.noise_in = args->noise_dev; D.noise_out = args->noise_out_dev;
    D.dataset = a->dataset_dev;
    D.dataset_bf16 = a->data_exact ? a->dataset_bf16_dev : nullptr; D.data_kp = a->data_kp;
    const bsvi_mlp_layer& Lloc = a->enc.layers[a->enc.producer[d.enc_loc_value]];
    const bsvi_mlp_layer& Lscale = a->enc.layers[a->enc.producer[d.enc_scale_value]];
    auto head_act = [](const bsvi_mlp_layer& L, uint32_t col, int& act, float& add) {
        const bool second = L.split_col > 0 && L.split_col < L.n_out && col >= L.split_col;
        act = (int)(second ? L.activation2 : L.activation);
        add = second ? L.post_add2 : L.post_add;
    };
    D.loc = val(a->enc, d.enc_loc_value) + d.enc_loc_col; D.ld_loc = a->enc.ld[d.enc_loc_value];
    D.scale = val(a->enc, d.enc_scale_value) + d.enc_scale_col; D.ld_scale = a->enc.ld[d.enc_scale_value];
    head_act(Lloc, d.enc_loc_col, D.act_loc, D.add_loc);
    head_act(Lscale, d.enc_scale_col, D.act_scale, D.add_scale);
    D.dloc = grad(a->enc, d.enc_loc_value) + d.enc_loc_col; D.dscale = grad(a->enc, d.enc_scale_value) + d.enc_scale_col;
    D.eps = eps; D.z = val(a->dec, 0); D.ld_z = a->dec.ld[0]; D.dz = grad(a->dec, 0);
    D.prior_loc = a->prior_dev; D.prior_scale = a->prior_dev + Dz;
    D.params = params; D.prior_loc_off = d.prior_loc_off; D.prior_scale_off = d.prior_scale_off;
    D.likelihood = (int)d.likelihood; D.lik_scale = a->lik_scale_dev;
    D.lik_scale_off = d.lik_scale_off; D.lik_scale_size = d.lik_scale_size; D.lik_scale_part = nullptr;
    D.lik_sd = nullptr; D.dlik_sd = nullptr;
    if (d.dec_scale_value) {
        D.lik_sd = val(a->dec, d.dec_scale_value); D.dlik_sd = grad(a->dec, d.dec_scale_value); D.ld_lik_sd = a->dec.ld[d.dec_scale_value];
        head_act(a->dec.layers[a->dec.producer[d.dec_scale_value]], 0, D.act_lik_sd, D.add_lik_sd);
    }
    D.rowf = rowf; D.rowlq = rowlq;
    D.logits = val(a->dec, d.dec_logits_value); D.ld_logits = a->dec.ld[d.dec_logits_value];
    D.out = out; D.fvalue_out = args->fvalue_out_dev; D.logq_out = args->logq_out_dev;
    if (!args->f_weight_dev != !args->q_weight_dev)
        return bsvi_fail(BSVI_ERR_INVALID, "f_weight_dev and q_weight_dev come together (the weights of grad f and grad log q per row), or not at all");
    D.f_weight = args->f_weight_dev; D.q_weight = args->q_weight_dev;

    const dim3 row_grid((unsigned)((R + 255) / 256));
    D.sum_part = part;
    part += align4(2 * (size_t)row_grid.x);
    if (int rc = add_segment(out, D.sum_part, 1, 2, row_grid.x)) return rc;       // out[0] = sum of values, out[1] = non-finite count
    if (d.prior_loc_off != BSVI_AMORT_CONSTANT || d.prior_scale_off != BSVI_AMORT_CONSTANT) {
        D.prior_loc_part = part;
        part += align4((size_t)row_grid.x * Dz);
        D.prior_scale_part = part;
        part += align4((size_t)row_grid.x * Dz);
        int rc = BSVI_OK;
        if (d.prior_loc_off != BSVI_AMORT_CONSTANT) rc = add_segment(grads + d.prior_loc_off, D.prior_loc_part, 1, (uint32_t)Dz, row_grid.x);
        if (!rc && d.prior_scale_off != BSVI_AMORT_CONSTANT) rc = add_segment(grads + d.prior_scale_off, D.prior_scale_part, 1, (uint32_t)Dz, row_grid.x);
        if (rc) return rc;
    }
    // the pieces of the weights the matrix-core products read — the wide layers' in both orientations (x6gemm_kernel), the data
    // layer's (xgemm_nt_kernel) — ride in the launch that draws the minibatch rows, when they fit its table of eight
    const bool x6_on = a->x6_any && R >= (size_t)kX6MinRows;
    bool data_pieces_done = false;
    {
        std::vector<X6SplitTable::Entry> entries;
        uint32_t blocks = 0;
        auto add = [&](const float* W, uint16_t* dst, int N, int K, int Kp, int transposed) {
            entries.push_back({W, dst, N, K, Kp, transposed, blocks});
            blocks += (uint32_t)(((size_t)N * Kp + 255) / 256);
        };
        if (a->data_exact)
            for (size_t l = 0; l < a->enc.layers.size() && l < a->weight_pieces.size(); ++l)
                if (a->enc.layers[l].in_value == 0 && a->weight_pieces[l])
                    add(params + a->enc.layers[l].weight_off, a->weight_pieces[l], (int)a->enc.layers[l].n_out, (int)a->enc.layers[l].n_in, a->data_kp, 0);
        const size_t n_data = entries.size();
        const Net* nets[2] = {&a->enc, &a->dec};
        for (int t = 0; t < 2 && x6_on; ++t)
            for (size_t l = 0; l < a->x6[t].size(); ++l) {
                const auto& X = a->x6[t][l];
                if (!X.nt) continue;
                const auto& L = nets[t]->layers[l];
                add(params + L.weight_off, X.nt, (int)L.n_out, (int)L.n_in, X.kp_nt, 0);
                add(params + L.weight_off, X.nn, (int)L.n_in, (int)L.n_out, X.kp_nn, 1);
            }
        if (!entries.empty() && entries.size() <= 8) {
            X6SplitTable T{};
            for (const auto& e : entries) T.e[T.n++] = e;
            hipLaunchKernelGGL(amort_head, dim3(row_grid.x + blocks), dim3(256), 0, stream, D, T, row_grid.x);
            data_pieces_done = n_data > 0;
        } else {
            hipLaunchKernelGGL(amort_rows, row_grid, dim3(256), 0, stream, D);
            // (a deeper network: the wide layers' pieces eight matrices per launch; the data layer's where it is multiplied)
            for (size_t at = n_data; at < entries.size(); at += 8) {
                X6SplitTable T{};
                const uint32_t first = entries[at].first_block;
                uint32_t last = blocks;
                for (size_t j = at; j < entries.size() && j < at + 8; ++j) { T.e[T.n] = entries[j]; T.e[T.n++].first_block -= first; }
                if (at + 8 < entries.size()) last = entries[at + 8].first_block;
                hipLaunchKernelGGL(x6_split_kernel, dim3(last - first), dim3(256), 0, stream, T);
            }
        }
        HIP_TRY(hipGetLastError());
    }

    // the transposed minibatch rows for the exact-data weight gradients (launched after the encoder's forward pass, below)
    uint16_t* XT = nullptr;
    const int Rp = (int)((R + 63) / 64 * 64);
    for (size_t li = 0; li < a->enc.layers.size() && !XT; ++li)
        if (a->enc.layers[li].in_value == 0 && xdw_layer(a, li, R)) {
            XT = reinterpret_cast<uint16_t*>(part);
            part += align4((size_t)P * Rp / 2);
        }

    // (round 6) the Bernoulli likelihood fused into the epilogue of the product that makes the logits (x6_epilogue, LIK): when that product
    // is a six-piece forward product without an activation, the data has its exact bf16 copy and no caller weights scale the seeds.
    // BSVI_AMORT_FUSE_LIK=0: the separate amort_lik launch (80 MB of logits out, in, and their gradient out again at config 5)
    const bool fuse_lik_env = [] { const char* e = getenv("BSVI_AMORT_FUSE_LIK"); return !(e && e[0] == '0'); }();      // (read per call: tests switch it)
    bool fuse_lik = false, lik_fused = false;
    float* lik_part = nullptr;
    {
        const int prod = a->dec.producer[d.dec_logits_value];
        const int x6m = [] { const char* e = getenv("BSVI_X6_MODES"); return e ? atoi(e) : 1; }();
        if (fuse_lik_env && prod >= 0 && d.likelihood == BSVI_AMORT_LIK_BINOMIAL1 && !args->f_weight_dev && a->data_exact && a->dataset_bf16_dev &&
            x6_on && (x6m & 1) && (size_t)prod < a->x6[1].size() && a->x6[1][prod].nt && (P & 3) == 0 && (a->data_kp & 3) == 0) {
            const auto& pl = a->dec.layers[prod];
            const bool plain = pl.activation == BSVI_ACT_NONE && pl.post_add == 0.0f && !(pl.split_col > 0 && pl.split_col < pl.n_out) && (int)pl.n_out == P;
            if (plain) {
                fuse_lik = true;
                lik_part = part;
                part += align4((size_t)((P + 127) / 128) * R);
                D.lik_part = lik_part; D.lik_tiles = (P + 127) / 128;
            }
        }
    }
    auto forward = [&](const Net& net, bool gather) -> int {
        for (size_t li = 0; li < net.layers.size(); ++li) {
            const auto& l = net.layers[li];
            GemmArgs G{};
            const bool from_data = gather && l.in_value == 0;
            if (from_data && a->data_exact && li < a->weight_pieces.size() && a->weight_pieces[li]) {
                // x W^T with x exact in bf16: three bf16 MFMAs per tile and step on the pieces of W (xgemm_nt_kernel)
                const int Kp = a->data_kp;
                uint16_t* pieces = a->weight_pieces[li];
                const long total = (long)l.n_out * Kp;
                if (!data_pieces_done)
                    hipLaunchKernelGGL(xw_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                                       params + l.weight_off, (int)l.n_out, (int)l.n_in, Kp, pieces);
                XGemmArgs X{};
                X.X = a->dataset_bf16_dev; X.rows = idx; X.Wp = pieces; X.plane_stride = total;
                X.C = val(net, l.out_value); X.ldc = net.ld[l.out_value];
                X.M = (int)R; X.N = (int)l.n_out; X.Kp = Kp;
                X.bias = l.bias_off != 0xFFFFFFFFu ? params + l.bias_off : nullptr;
                X.act = (int)l.activation; X.post_add = l.post_add;
                if (l.split_col > 0 && l.split_col < l.n_out) { X.split = (int)l.split_col; X.act2 = (int)l.activation2; X.post_add2 = l.post_add2; }
                launch_xgemm(X, stream);
                HIP_TRY(hipGetLastError());
                continue;
            }
            const auto& x6 = a->x6[&net == &a->dec ? 1 : 0];
            const int x6_modes = [] { const char* e = getenv("BSVI_X6_MODES"); return e ? atoi(e) : 1; }();      // (bit 0: forward, bit 1: input gradient — see backward; read per call: tests switch it)
            if (x6_on && (x6_modes & 1) && !from_data && li < x6.size() && x6[li].nt) {
                X6Args X{};
                X.A = val(net, l.in_value); X.lda = net.ld[l.in_value];
                X.Bp = x6[li].nt; X.Kp = x6[li].kp_nt; X.plane_stride = (long)l.n_out * X.Kp;
                X.C = val(net, l.out_value); X.ldc = net.ld[l.out_value];
                X.M = (int)R; X.N = (int)l.n_out; X.K = (int)l.n_in;
                X.bias = l.bias_off != 0xFFFFFFFFu ? params + l.bias_off : nullptr;
                X.act = (int)l.activation; X.post_add = l.post_add;
                if (l.split_col > 0 && l.split_col < l.n_out) { X.split = (int)l.split_col; X.act2 = (int)l.activation2; X.post_add2 = l.post_add2; }
                if (fuse_lik && &net == &a->dec && l.out_value == d.dec_logits_value) {
                    // the Bernoulli likelihood in this product's epilogue: d f / d logits where the logits would have gone (the buffers are
                    // one: the logits are overwritten by their own gradient), one partial log-likelihood per (row, column tile)
                    X.lik_x = a->dataset_bf16_dev; X.lik_kp = a->data_kp; X.lik_idx = idx; X.lik_part = lik_part;
                    lik_fused = true;
                }
                if (int rc = launch_x6(false, X, stream)) return rc;
                continue;
            }
            G.A = from_data ? a->dataset_dev : val(net, l.in_value);
            G.lda = from_data ? P : net.ld[l.in_value];
            G.rows = from_data ? idx : nullptr;
            G.B = params + l.weight_off; G.ldb = (int)l.n_in;
            G.C = val(net, l.out_value); G.ldc = net.ld[l.out_value];
            G.M = (int)R; G.N = (int)l.n_out; G.K = (int)l.n_in;
            G.bias = l.bias_off != 0xFFFFFFFFu ? params + l.bias_off : nullptr;
            G.act = (int)l.activation; G.post_add = l.post_add;
            if (l.split_col > 0 && l.split_col < l.n_out) { G.split = (int)l.split_col; G.act2 = (int)l.activation2; G.post_add2 = l.post_add2; }
            int rc = launch_gemm(MODE_NT, G, stream);
            if (rc) return rc;
        }
        return BSVI_OK;
    };
    size_t next_event = 0;
    // (round 5) the data layer's weight gradient through x6tn_kernel<., true>: no transposed minibatch rows, no pieces of dY in memory.  BSVI_X6_TN_DATA=0: round 4's three launches
    static const bool data_x6tn_env = [] { const char* e = getenv("BSVI_X6_TN_DATA"); return !(e && e[0] == '0'); }();
    static const bool x6_tn_env = [] { const char* e = getenv("BSVI_X6_TN"); return !(e && e[0] == '0'); }();
    const bool data_x6tn = x6_on && x6_tn_env && data_x6tn_env;
    bool xt_pending = XT != nullptr && !data_x6tn, xt_on_side = false;
    auto launch_xt = [&](hipStream_t ts) {
        hipLaunchKernelGGL(xt_gather_kernel, dim3((unsigned)(Rp / 64), (unsigned)((P + 127) / 128)), dim3(256), 0, ts,
                           a->dataset_bf16_dev, idx, a->data_kp, P, (int)R, Rp, XT);
    };
    auto backward = [&](const Net& net, bool gather, bool input_grad) -> int {
        std::vector<char> written(net.width.size(), 0);
        for (int i = (int)net.layers.size() - 1; i >= 0; --i) {
            hipStream_t wstream = stream;
            const auto& lw = net.layers[i];
            // (a layer without an input-gradient product — the one that reads the data rows — has nothing on `stream` to
            //  run beside: its weight gradient stays there and saves the event round trip, ~7 us)
            if (a->overlap && (lw.in_value != 0 || input_grad)) {          // dY of this layer is complete on `stream` here
                hipEvent_t e = a->ready[next_event++];
                HIP_TRY(hipEventRecord(e, stream));
                HIP_TRY(hipStreamWaitEvent(a->side, e, 0));
                wstream = a->side;
                if (xt_pending) {
                    // the transposed minibatch rows, first thing on the side stream of the backward pass: the side stream has
                    // ~70 us of slack there (beside the encoder's first product it cost that product 12 us, beside the
                    // decoder's narrow first layer 16 us and an event round trip of its own)
                    launch_xt(a->side);
                    HIP_TRY(hipEventRecord(a->ready.back(), a->side));       // (the exact-piece weight gradient runs on `stream`)
                    xt_on_side = true;
                    xt_pending = false;
                }
            }
            const auto& l = net.layers[i];
            const float* dY = grad(net, l.out_value);
            const int ldy = net.ld[l.out_value];
            const bool from_data = gather && l.in_value == 0;
            if (from_data && XT && xdw_layer(a, (size_t)i, R) && data_x6tn) {
                // (round 5) the same product through x6tn_kernel: dY split and transposed on the way into LDS, the data rows gathered by
                // the kernel and converted (they are exactly bf16: one piece, three products) — no transposed copies in memory, one launch
                // in place of xt_gather + dy_split_t + the exact-piece product
                const X6TnPlan xp = x6tn_plan((int)l.n_out, (int)l.n_in, (int)R);
                const bool has_bias = l.bias_off != 0xFFFFFFFFu;
                float* const cpart = part;
                float* const bpart = cpart + align4((size_t)xp.slices * l.n_out * l.n_in);
                part += xdw_floats(a, (size_t)i, R);
                X6TnArgs X{};
                X.A = dY; X.lda = ldy; X.B = a->dataset_dev; X.ldb = (int)P; X.rows = idx; X.C = cpart; X.part_stride = (long)l.n_out * l.n_in;
                X.bias_grad = has_bias ? bpart : nullptr;
                X.M = (int)l.n_out; X.N = (int)l.n_in; X.K = (int)R; X.chunk = xp.chunk; X.tiles = xp.tiles;
                hipLaunchKernelGGL((x6tn_kernel<0, true>), dim3((unsigned)(xp.tiles * xp.slices)), dim3(256), 0, wstream, X);
                HIP_TRY(hipGetLastError());
                int rc = add_segment(grads + l.weight_off, cpart, l.n_out, l.n_in, (uint32_t)xp.slices);
                if (!rc && has_bias) rc = add_segment(grads + l.bias_off, bpart, 1, l.n_out, (uint32_t)xp.slices);
                if (rc) return rc;
            } else
            if (from_data && XT && xdw_layer(a, (size_t)i, R)) {
                // dW = dY^T x with x exact in bf16: the pieces of dY, transposed, times the transposed rows (split-k)
                const XdwPlan plan = xdw_plan((int)l.n_in, (int)l.n_out, R);
                const bool has_bias = l.bias_off != 0xFFFFFFFFu;
                float* slices = part;
                part += align4((size_t)plan.slices * l.n_out * l.n_in);
                float* colsum = part;
                part += align4((size_t)(Rp / 64) * l.n_out);
                uint16_t* T = reinterpret_cast<uint16_t*>(part);
                part += align4(3 * (size_t)l.n_out * Rp / 2);
                if (xt_on_side && wstream != a->side) HIP_TRY(hipStreamWaitEvent(wstream, a->ready.back(), 0));
                // (round 4, measured and removed: the pieces and column sums written by the EPILOGUE of the input-gradient product that
                //  computes dY — 64 x 128 tile through LDS, no dy_split_t launch, dY not read back: 1.0276 against 1.0278 ms per cfg 5
                //  iteration.  The end of the backward pass is bound by the work of both streams together — that product grew from
                //  118 to 133 us beside the side stream's weight gradient — not by the length of this stream's chain.)
                hipLaunchKernelGGL(dy_split_t_kernel, dim3((unsigned)(Rp / 64), (unsigned)((l.n_out + 63) / 64)), dim3(256), 0, wstream,
                                   dY, ldy, (int)R, Rp, (int)l.n_out, T, has_bias ? colsum : nullptr);
                launch_xdw(XT, T, (int)l.n_in, (int)l.n_out, plan, slices, wstream);
                HIP_TRY(hipGetLastError());
                int rc = add_segment(grads + l.weight_off, slices, l.n_out, l.n_in, (uint32_t)plan.slices);
                if (!rc && has_bias) rc = add_segment(grads + l.bias_off, colsum, 1, l.n_out, (uint32_t)(Rp / 64));
                if (rc) return rc;
            } else
            {   // dW[n_out][n_in] = dY^T x, db = 1^T dY (same launch): one partial per slice of the rows
                GemmArgs G{};
                G.A = dY; G.lda = ldy;
                G.B = from_data ? a->dataset_dev : val(net, l.in_value);
                G.ldb = from_data ? P : net.ld[l.in_value];
                G.rows = from_data ? idx : nullptr;
                G.M = (int)l.n_out; G.N = (int)l.n_in; G.K = (int)R;
                const TnPlan plan = tn_plan(G.M, G.N, G.K);
                const bool has_bias = l.bias_off != 0xFFFFFFFFu;
                G.C = part; G.ldc = (int)l.n_in; G.part_stride = (long)G.M * G.N;
                part += align4((size_t)plan.slices * G.M * G.N);
                G.bias_grad = has_bias ? part : nullptr;
                if (has_bias) part += align4((size_t)plan.slices * G.M);
                // wide layers whose input is a network value: six products of exact pieces with the transposing split of both operands
                // on the way into LDS (x6tn_kernel; round 5) — 6 / 16 of the f32-input MFMA's pipe time.  BSVI_X6_TN=0: the f32-input kernel
                static const bool x6_tn = [] { const char* e = getenv("BSVI_X6_TN"); return !(e && e[0] == '0'); }();
                if (x6_on && x6_tn && !from_data && !plan.skinny && G.M >= 64 && G.N >= 64) {
                    const X6TnPlan xp = x6tn_plan(G.M, G.N, G.K);
                    // (the partial regions were laid out for the larger of the two plans: tn_partial_floats)
                    float* const cpart = G.C;
                    float* bpart = nullptr;
                    part = cpart + align4((size_t)std::max(plan.slices, xp.slices) * G.M * G.N);
                    if (has_bias) { bpart = part; part += align4((size_t)std::max(plan.slices, xp.slices) * G.M); }
                    X6TnArgs X{};
                    X.A = G.A; X.lda = G.lda; X.B = G.B; X.ldb = G.ldb; X.C = cpart; X.part_stride = (long)G.M * G.N; X.bias_grad = bpart;
                    X.M = G.M; X.N = G.N; X.K = G.K; X.chunk = xp.chunk; X.tiles = xp.tiles;
                    hipLaunchKernelGGL((x6tn_kernel<0>), dim3((unsigned)(xp.tiles * xp.slices)), dim3(256), 0, wstream, X);
                    HIP_TRY(hipGetLastError());
                    int rc = add_segment(grads + l.weight_off, cpart, l.n_out, l.n_in, (uint32_t)xp.slices);
                    if (!rc && has_bias) rc = add_segment(grads + l.bias_off, bpart, 1, l.n_out, (uint32_t)xp.slices);
                    if (rc) return rc;
                } else {
                int rc = launch_gemm(MODE_TN, G, wstream);
                if (!rc) rc = add_segment(grads + l.weight_off, G.C, l.n_out, l.n_in, (uint32_t)plan.slices);
                if (!rc && has_bias) rc = add_segment(grads + l.bias_off, G.bias_grad, 1, l.n_out, (uint32_t)plan.slices);
                if (rc) return rc;
                }
            }
            if (l.in_value != 0 || input_grad) {   // dX = (dY W) * act'(x)
                GemmArgs G{};
                G.A = dY; G.lda = ldy;
                G.B = params + l.weight_off; G.ldb = (int)l.n_in;
                G.C = grad(net, l.in_value); G.ldc = net.ld[l.in_value];
                G.M = (int)R; G.N = (int)l.n_in; G.K = (int)l.n_out;
                const int prod = net.producer[l.in_value];
                if (prod >= 0) {
                    const auto& pl = net.layers[prod];
                    G.act = (int)pl.activation; G.post_add = pl.post_add;
                    if (pl.split_col > 0 && pl.split_col < pl.n_out) { G.split = (int)pl.split_col; G.act2 = (int)pl.activation2; G.post_add2 = pl.post_add2; }
                    const bool any_act = G.act != BSVI_ACT_NONE || (G.split > 0 && G.act2 != BSVI_ACT_NONE);
                    if (any_act) { G.Y = val(net, l.in_value); G.ldy = net.ld[l.in_value]; }
                }
                G.accumulate = written[l.in_value];
                written[l.in_value] = 1;
                const auto& x6 = a->x6[&net == &a->dec ? 1 : 0];
                // Round 6: EVERY wide input gradient runs as six products of exact pieces (x6gemm_kernel<true>).  Rounds 4 / 5 kept the ones
                // contracting fewer than 512 columns on the f32-input MFMA kernel: with the register-staged kernel a tile's prologue / epilogue
                // weighed against 8 k steps lost (cfg 5, one at a time beside the f32-input weight gradients: K = 784 -> 0.976 ms, 512 -> 0.984,
                // 256 -> 1.020 against 0.984 with none).  With the LDS-DMA kernel all three win: cfg 5 0.800 -> 0.769 ms with the K = 256 one
                // on it too (profiles/r6/x6_notes.txt).  BSVI_X6_MODES=1: none (the f32-input kernel), 3: all.
                // (What trying them in round 4 found is why this file is compiled without packed-f32 instructions — Makefile: with
                //  x6gemm_kernel<true> on this stream the narrow layers' outer_kernel on the other one returned ~50 of 1024 values different
                //  from call to call, lanes 48-63, beside EVERY bf16-MFMA kernel: tools/r4/coresidency_probe.py, profiles/r4/x6_notes.txt 4.)
                const int x6_modes = [&] { const char* e = getenv("BSVI_X6_MODES"); return e ? atoi(e) : 3; }();
                static const int x6_only = [] { const char* e = getenv("BSVI_X6_NN_ONLY"); return e ? atoi(e) : -1; }();     // (diagnostics: 10 * net + layer)
                if (x6_on && (x6_modes & 2) && (size_t)i < x6.size() && x6[i].nn && (x6_only < 0 || x6_only == 10 * (&net == &a->dec ? 1 : 0) + i)) {
                    X6Args X{};
                    X.A = dY; X.lda = ldy;
                    X.Bp = x6[i].nn; X.Kp = x6[i].kp_nn; X.plane_stride = (long)l.n_in * X.Kp;
                    X.C = G.C; X.ldc = G.ldc; X.M = G.M; X.N = G.N; X.K = G.K;
                    X.Y = G.Y; X.ldy = G.ldy; X.act = G.act; X.post_add = G.post_add;
                    X.split = G.split; X.act2 = G.act2; X.post_add2 = G.post_add2; X.accumulate = G.accumulate;
                    if (int rc = launch_x6(true, X, stream)) return rc;
                    continue;
                }
                int rc = launch_gemm(MODE_NN, G, stream);
                if (rc) return rc;
            }
        }
        return BSVI_OK;
    };

    int rc = forward(a->enc, true);
    if (rc) return rc;
    hipLaunchKernelGGL(amort_latent_fwd, row_grid, dim3(256), 0, stream, D);
    rc = forward(a->dec, false);
    if (rc) return rc;
    if (!lik_fused) {
        D.lik_part = nullptr; D.lik_tiles = 0;                // (the product fell back to another kernel: the separate launch)
        hipLaunchKernelGGL(amort_lik, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, stream, D);
    }
    if (d.lik_scale_off != BSVI_AMORT_CONSTANT) {
        // d log p / d (raw scale): column sums over the d log p / d mean that amort_lik left, per slice of the rows
        const uint32_t slices = lik_scale_slices(R);
        const int per = (int)((R + slices - 1) / slices);
        D.lik_scale_part = part;
        part += align4((size_t)slices * P);
        hipLaunchKernelGGL(amort_lik_scale_grad, dim3((unsigned)((P + 255) / 256), slices), dim3(256), 0, stream, D, per);
        // one value for every feature: its gradient is the sum over the features too (slices x P partials of ONE element)
        rc = d.lik_scale_size == 1u ? add_segment(grads + d.lik_scale_off, D.lik_scale_part, 1, 1, slices * (uint32_t)P)
                                    : add_segment(grads + d.lik_scale_off, D.lik_scale_part, 1, (uint32_t)P, slices);
        if (rc) return rc;
    }
    rc = backward(a->dec, false, true);
    if (rc) return rc;
    // several ranks: the decoder's partial sums are reduced NOW, on the host's bucket stream (behind the kernels that write them, on
    // either stream), so that its all-reduce can run beside the encoder's backward pass
    uint64_t early_mask = 0;                    // segments reduced on the bucket stream (those of the decoder's range that exist by now)
    if (a->bucket_stream && a->bucket_count) {
        SegmentTable early{};
        uint32_t early_blocks = 0;
        for (int k = 0; k < segments.n && k < 64; ++k) {
            const float* const dst = segments.seg[k].dst;
            if (dst < grads + a->bucket_first || dst >= grads + a->bucket_first + a->bucket_count) continue;
            Segment& S = early.seg[early.n++];
            S = segments.seg[k];
            S.first_block = early_blocks;
            early_blocks += segment_blocks(S);
            early_mask |= 1ull << k;
        }
        if (early.n) {
            HIP_TRY(hipEventRecord(a->bucket_ev[0], stream));
            HIP_TRY(hipStreamWaitEvent(a->bucket_stream, a->bucket_ev[0], 0));
            if (a->overlap) {
                HIP_TRY(hipEventRecord(a->bucket_ev[1], a->side));
                HIP_TRY(hipStreamWaitEvent(a->bucket_stream, a->bucket_ev[1], 0));
            }
            hipLaunchKernelGGL(reduce_partials, dim3(early_blocks), dim3(256), 0, a->bucket_stream, early);
        }
    }
    hipLaunchKernelGGL(amort_latent_bwd, row_grid, dim3(256), 0, stream, D);
    if (xt_pending) { launch_xt(stream); xt_pending = false; }       // (no side stream: BSVI_AMORT_OVERLAP=0)
    rc = backward(a->enc, true, false);
    if (rc) return rc;
    if (a->overlap) {
        HIP_TRY(hipEventRecord(a->joined, a->side));
        HIP_TRY(hipStreamWaitEvent(stream, a->joined, 0));
    }
    if (early_mask) {                           // (the decoder's segments went out on the bucket stream)
        SegmentTable rest{};
        uint32_t rest_blocks = 0;
        for (int k = 0; k < segments.n; ++k) {
            if (k < 64 && ((early_mask >> k) & 1ull)) continue;
            Segment& S = rest.seg[rest.n++];
            S = segments.seg[k];
            S.first_block = rest_blocks;
            rest_blocks += segment_blocks(S);
        }
        if (rest.n) hipLaunchKernelGGL(reduce_partials, dim3(rest_blocks), dim3(256), 0, stream, rest);
    } else {
        hipLaunchKernelGGL(reduce_partials, dim3(reduce_blocks), dim3(256), 0, stream, segments);
    }
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}

// the decoder's parameters as one range of the parameter vector (count 0: they are not one), and the stream on which their
// gradients are to be complete early (NULL: everything is reduced by one launch at the end of bsvi_amort_fwd_bwd, the default)
extern "C" int bsvi_amort_bucket(const bsvi_amort* a, uint32_t* first_param, uint32_t* n_params) {
    if (!a || !first_param || !n_params) return bsvi_fail(BSVI_ERR_INVALID, "null argument");
    *first_param = a->bucket_first;
    *n_params = a->bucket_count;
    return BSVI_OK;
}
extern "C" int bsvi_amort_set_bucket_stream(bsvi_amort* a, void* stream) {
    if (!a) return bsvi_fail(BSVI_ERR_INVALID, "null argument");
    if (stream && !a->bucket_count) return bsvi_fail(BSVI_ERR_UNSUPPORTED, "the decoder's parameters are not one range of the parameter vector");
    for (auto& e : a->bucket_ev)
        if (stream && !e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, "hipEventCreate");
    a->bucket_stream = (hipStream_t)stream;
    return BSVI_OK;
}

// Forward pass of ONE network on caller-supplied rows (posterior predictive / encoding: the step after training in
// examples/VAE_playground.py:90-103 — `model.get_sample(1, input_values={z: ...})["decoder_output"]`).
extern "C" int bsvi_amort_apply(const bsvi_amort* a, int network, const float* params_dev, const float* input_dev,
                                uint32_t n_rows, uint32_t value, float* out_dev, void* workspace_dev, void* stream_) {
    if (!a || !params_dev || !input_dev || !out_dev || !workspace_dev || !n_rows) return bsvi_fail(BSVI_ERR_INVALID, "null argument");
    if (network != 0 && network != 1) return bsvi_fail(BSVI_ERR_INVALID, "network must be 0 (encoder) or 1 (decoder)");
    const Net& net = network == 0 ? a->enc : a->dec;
    if (value == 0 || value >= net.width.size()) return bsvi_fail(BSVI_ERR_INVALID, "no such network value");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t R = n_rows;
    if (R * a->floats_per_row > bsvi_amort_workspace_bytes(a, (n_rows + a->d.batch_size - 1) / a->d.batch_size) / sizeof(float))
        return bsvi_fail(BSVI_ERR_INVALID, "workspace too small for this many rows");
    float* ws = (float*)workspace_dev;
    // private layout of this call: value v at offsets accumulated over the network's values
    std::vector<size_t> off(net.width.size());
    size_t acc = 0;
    for (size_t v = 0; v < net.width.size(); ++v) { off[v] = acc; acc += (size_t)net.ld[v]; }
    auto val = [&](uint32_t v) { return ws + off[v] * R; };
    {
        const long total = (long)R * net.width[0];
        hipLaunchKernelGGL(copy_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, input_dev, net.width[0],
                           val(0), net.ld[0], (long)R, net.width[0]);
    }
    for (const auto& l : net.layers) {
        GemmArgs G{};
        G.A = val(l.in_value); G.lda = net.ld[l.in_value];
        G.B = params_dev + l.weight_off; G.ldb = (int)l.n_in;
        G.C = val(l.out_value); G.ldc = net.ld[l.out_value];
        G.M = (int)R; G.N = (int)l.n_out; G.K = (int)l.n_in;
        G.bias = l.bias_off != 0xFFFFFFFFu ? params_dev + l.bias_off : nullptr;
        G.act = (int)l.activation; G.post_add = l.post_add;
        if (l.split_col > 0 && l.split_col < l.n_out) { G.split = (int)l.split_col; G.act2 = (int)l.activation2; G.post_add2 = l.post_add2; }
        int rc = launch_gemm(MODE_NT, G, stream);
        if (rc) return rc;
    }
    const long total = (long)R * net.width[value];
    hipLaunchKernelGGL(copy_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, val(value), net.ld[value],
                       out_dev, net.width[value], (long)R, net.width[value]);
    HIP_TRY(hipGetLastError());
    return BSVI_OK;
}
