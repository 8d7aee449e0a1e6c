// bsvi_device.h — device helpers shared by the interpreter kernels (elbo_kernel.hip), the dense / amortised
// kernels and the program-specialised kernels that libbsvi generates at run time (specialize.cpp; this header is
// embedded in the library and handed to hiprtc together with dist_math.h, philox.h and include/bsvi.h).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#include "bsvi.h"
#include "dist_math.h"
#include "philox.h"

namespace bsvi {

// ---------------------------------------------------------------------------------------
// wave-level sum of one float per lane; result is wave-uniform (every lane gets it).
// DPP row reduction (quad_perm, row_half_mirror, row_mirror) then 4 row totals through
// v_readlane — fixed order, no LDS traffic.
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);   // row_half_mirror
    v += dpp_f<0x140>(v);   // row_mirror  -> every lane holds its 16-lane row total
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ float utransform(int t, float x) {
    switch (t) {
    case BSVI_UT_SOFTPLUS: return softplusf_(x);
    case BSVI_UT_SIGMOID: return sigmoidf_(x);
    case BSVI_UT_EXP: return expf(x);
    case BSVI_UT_LOG: return logf(x);
    case BSVI_UT_TANH: return tanhf(x);
    case BSVI_UT_SQRT: return sqrtf(x);
    case BSVI_UT_SQUARE: return x * x;
    default: return x;
    }
}
__device__ __forceinline__ float utransform_grad(int t, float x) {
    switch (t) {
    case BSVI_UT_SOFTPLUS: return x > 20.0f ? 1.0f : sigmoidf_(x);
    case BSVI_UT_SIGMOID: { const float s = sigmoidf_(x); return s * (1.0f - s); }
    case BSVI_UT_EXP: return expf(x);
    case BSVI_UT_LOG: return 1.0f / x;
    case BSVI_UT_TANH: { const float th = tanhf(x); return 1.0f - th * th; }
    case BSVI_UT_SQRT: return 0.5f / sqrtf(x);
    case BSVI_UT_SQUARE: return 2.0f * x;
    default: return 1.0f;
    }
}

struct PhiloxKey { uint32_t nidx, seed_lo, seed_hi, off_lo, off_hi; };

// Counter words: c0 = global sample index, c1 = noise row | attempt << 16 (bit 31 is the flag of the Normal draws, which
// use c1 = group | 0x80000000), c2 / c3 = the 64-bit offset (iteration), untouched — so no retry of a rejection sampler and
// no block of a Binomial can land on the stream of another iteration, whatever the offset (sampling programs run at
// offsets >= 2^62).  A noise row is a slot of the lane's row in LDS (<= 160 KB / 8 B = 20 480 of them): it fits 16 bits;
// attempts are (stream << 12) + retry for the gamma sampler and total_count / 4 for a Binomial: 15 bits.
__device__ __forceinline__ u32x4 philox_raw(const PhiloxKey& k, uint32_t row, uint32_t attempt) {
    return philox4x32(k.nidx, (row & 0xFFFFu) | ((attempt & 0x7FFFu) << 16), k.off_lo, k.off_hi, k.seed_lo, k.seed_hi);
}
// Box-Muller on the hardware transcendental units: v_sin/v_cos take revolutions, so
// sin(2*pi*u) is one instruction and needs no range reduction
__device__ __forceinline__ void box_muller_fast(uint32_t a, uint32_t b, float& z0, float& z1) {
    // u01 is a normal float in (0, 1): v_log_f32 (log2) and v_sqrt_f32 need no denormal or range handling — the
    // library logf / sqrtf spend ~20 instructions per pair on exactly that
    const float r = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u01(a)) * -1.3862943611198906f);     // sqrt(-2 ln u)
    const float u = u01(b);
    z0 = r * __builtin_amdgcn_cosf(u);
    z1 = r * __builtin_amdgcn_sinf(u);
}
__device__ __noinline__ float philox_gamma(PhiloxKey G, float alpha, uint32_t row, uint32_t stream) {
    // Marsaglia & Tsang (2000), as ATen/native/Distributions.h sample_gamma
    float scale = 1.0f;
    uint32_t attempt = stream << 12;
    if (alpha < 1.0f) {
        if (alpha == 0.0f) return 0.0f;
        const u32x4 x = philox_raw(G, row, attempt++);
        scale *= powf(1.0f - u01(x.x), 1.0f / alpha);
        alpha += 1.0f;
    }
    const float d = alpha - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
    for (int it = 0; it < 64; ++it) {
        const u32x4 x = philox_raw(G, row, attempt++);
        float n0, n1;
        box_muller(x.x, x.y, n0, n1);
        const float y = 1.0f + c * n0;
        if (y <= 0.0f) continue;
        const float v = y * y * y, u = 1.0f - u01(x.z), xx = n0 * n0;
        if (u < 1.0f - 0.0331f * xx * xx) return scale * d * v;
        if (logf(u) < 0.5f * xx + d * (1.0f - v + logf(v))) return scale * d * v;
    }
    return scale * d;
}

// a fresh draw for the non-Normal distributions (out of line: cold for the AR-type models);
// returns the draw in .x and the base noise that produced it in .y
BSVI_SWITCH_FN float2 philox_draw(PhiloxKey G, int dist, float p0, float p1, uint32_t row) {
    float noise = 0.0f, v = p0;
    switch (dist) {
    case BSVI_DIST_LOGNORMAL: {
        const u32x4 x = philox_raw(G, row, 0);
        float n1;
        box_muller(x.x, x.y, noise, n1);
        v = expf(p0 + noise * p1);
        break;
    }
    case BSVI_DIST_CAUCHY: {
        const u32x4 x = philox_raw(G, row, 0);
        noise = tanf(3.14159265358979323846f * (u01(x.x) - 0.5f));
        v = p0 + noise * p1;
        break;
    }
    case BSVI_DIST_LAPLACE: {
        const u32x4 x = philox_raw(G, row, 0);
        noise = (kFloatEps - 1.0f) + (2.0f - kFloatEps) * u01(x.x);   // torch laplace.py:83
        v = sample_from_noise_generic(dist, p0, p1, noise);
        break;
    }
    case BSVI_DIST_BETA: {
        const float ga = philox_gamma(G, p0, row, 1), gb = philox_gamma(G, p1, row, 2);
        v = noise = fminf(fmaxf(ga / (ga + gb), 1.17549435e-38f), 1.0f - kFloatEps);
        break;
    }
    case BSVI_DIST_BERNOULLI: {
        const u32x4 x = philox_raw(G, row, 0);
        v = noise = u01(x.x) < sigmoidf_(p0) ? 1.0f : 0.0f;
        break;
    }
    case BSVI_DIST_BINOMIAL: {
        const float p = sigmoidf_(p1);
        const int n = (int)p0;
        float k = 0.0f;
        for (int i = 0; i < n; i += 4) {
            const u32x4 x = philox_raw(G, row, (uint32_t)(i >> 2));
            k += (u01(x.x) < p) ? 1.0f : 0.0f;
            if (i + 1 < n) k += (u01(x.y) < p) ? 1.0f : 0.0f;
            if (i + 2 < n) k += (u01(x.z) < p) ? 1.0f : 0.0f;
            if (i + 3 < n) k += (u01(x.w) < p) ? 1.0f : 0.0f;
        }
        v = noise = k;
        break;
    }
    default: break;
    }
    return make_float2(v, noise);
}
// the base noise again, for the reverse sweep of a reparameterised non-Normal draw
BSVI_SWITCH_FN float philox_noise_again(PhiloxKey G, int dist, uint32_t row) {
    const u32x4 x = philox_raw(G, row, 0);
    if (dist == BSVI_DIST_LOGNORMAL) { float n0, n1; box_muller(x.x, x.y, n0, n1); return n0; }
    if (dist == BSVI_DIST_CAUCHY) return tanf(3.14159265358979323846f * (u01(x.x) - 0.5f));
    if (dist == BSVI_DIST_LAPLACE) return (kFloatEps - 1.0f) + (2.0f - kFloatEps) * u01(x.x);
    return 0.0f;
}

// ---------------------------------------------------------------------------------------
// optimizer arithmetic shared by reduce_kernel / optimizer_kernel / persistent trainer
//   torch.optim.SGD / torch.optim.Adam single-tensor paths, per element.
// state layout: [4][n_params] = (momentum_buffer | exp_avg, exp_avg_sq, max_exp_avg_sq, step)
// ---------------------------------------------------------------------------------------
// on values (registers): p = parameter, (s0, s1, s2, st) = its four state words
__device__ __forceinline__ void optimizer_apply(const bsvi_opt_cfg& cfg, float& p, float& s0, float& s1, float& s2, float& st,
                                                float grad) {
    if (cfg.maximize) grad = -grad;
    const float step = st + 1.0f;
    st = step;
    if (cfg.kind == BSVI_OPT_SGD) {
        if (cfg.weight_decay != 0.0f) grad += cfg.weight_decay * p;
        if (cfg.momentum != 0.0f) {
            const float buf = (step == 1.0f) ? grad : cfg.momentum * s0 + (1.0f - cfg.dampening) * grad;
            s0 = buf;
            grad = cfg.nesterov ? grad + cfg.momentum * buf : buf;
        }
        p = p - cfg.lr * grad;
    } else {
        if (cfg.weight_decay != 0.0f) grad += cfg.weight_decay * p;
        const float m = s0 + (grad - s0) * (1.0f - cfg.beta1);            // exp_avg.lerp_(grad, 1 - beta1)
        const float v = cfg.beta2 * s1 + (1.0f - cfg.beta2) * grad * grad;
        s0 = m;
        s1 = v;
        const double bc1 = 1.0 - pow((double)cfg.beta1, (double)step);
        const double bc2 = 1.0 - pow((double)cfg.beta2, (double)step);
        const float step_size = (float)((double)cfg.lr / bc1);
        const float bc2_sqrt = (float)sqrt(bc2);
        float vhat = v;
        if (cfg.amsgrad) {
            vhat = fmaxf(s2, v);
            s2 = vhat;
        }
        const float denom = sqrtf(vhat) / bc2_sqrt + cfg.eps;
        p = p - step_size * (m / denom);
    }
}
// The same step for a caller that keeps beta1^st and beta2^st as running double-precision products (p1, p2) next to the
// state — the in-kernel training loop: one multiply per step instead of a double-precision pow (two of them were a fifth
// of an iteration of BASELINE config 1 under Adam).  The products differ from pow() by the rounding of at most `st`
// multiplies (~st * 1e-16 relative): far below the single-precision rounding of the step size they end up in.
__device__ __forceinline__ void optimizer_apply_running(const bsvi_opt_cfg& cfg, float& p, float& s0, float& s1, float& s2, float& st,
                                                        float grad, double& p1, double& p2) {
    if (cfg.kind == BSVI_OPT_SGD) { optimizer_apply(cfg, p, s0, s1, s2, st, grad); return; }
    if (cfg.maximize) grad = -grad;
    st = st + 1.0f;
    p1 *= (double)cfg.beta1;
    p2 *= (double)cfg.beta2;
    if (cfg.weight_decay != 0.0f) grad += cfg.weight_decay * p;
    const float m = s0 + (grad - s0) * (1.0f - cfg.beta1);
    const float v = cfg.beta2 * s1 + (1.0f - cfg.beta2) * grad * grad;
    s0 = m;
    s1 = v;
    const float step_size = (float)((double)cfg.lr / (1.0 - p1));
    const float bc2_sqrt = (float)sqrt(1.0 - p2);
    float vhat = v;
    if (cfg.amsgrad) {
        vhat = fmaxf(s2, v);
        s2 = vhat;
    }
    const float denom = sqrtf(vhat) / bc2_sqrt + cfg.eps;
    p = p - step_size * (m / denom);
}
// on memory: state planes `n_params` words apart
__device__ __forceinline__ void optimizer_update(const bsvi_opt_cfg& cfg, float* params, float* state,
                                                 uint32_t n_params, uint32_t i, float grad) {
    float p = params[i], s0 = state[i], s1 = state[n_params + i], s2 = state[2 * (size_t)n_params + i];
    float st = state[3 * (size_t)n_params + i];
    optimizer_apply(cfg, p, s0, s1, s2, st, grad);
    params[i] = p;
    state[i] = s0;
    state[n_params + i] = s1;
    state[2 * (size_t)n_params + i] = s2;
    state[3 * (size_t)n_params + i] = st;
}

// ---- unary link functions (brancher/functions.py:28-41) and their derivatives; GEN=false compiles the rare ones out
__device__ __noinline__ float unop_rare(uint32_t sub, float x, float imm) {
    switch (sub) {
    case BSVI_U_SIN: return sinf(x);
    case BSVI_U_COS: return cosf(x);
    case BSVI_U_TANH: return tanhf(x);
    case BSVI_U_LOG1P: return log1pf(x);
    case BSVI_U_EXPM1: return expm1f(x);
    case BSVI_U_P2L: { const float p = fminf(fmaxf(x, kFloatEps), 1.0f - kFloatEps); return logf(p) - log1pf(-p); }
    case BSVI_U_POWI: return powf(x, imm);
    default: return x;
    }
}
__device__ __noinline__ float unop_rare_grad(uint32_t sub, float x, float y, float imm) {
    switch (sub) {
    case BSVI_U_SIN: return cosf(x);
    case BSVI_U_COS: return -sinf(x);
    case BSVI_U_TANH: return 1.0f - y * y;
    case BSVI_U_LOG1P: return 1.0f / (1.0f + x);
    case BSVI_U_EXPM1: return y + 1.0f;
    case BSVI_U_P2L: return (x >= kFloatEps && x <= 1.0f - kFloatEps) ? (1.0f / x + 1.0f / (1.0f - x)) : 0.0f;
    case BSVI_U_POWI: return imm * powf(x, imm - 1.0f);
    default: return 1.0f;
    }
}
template <bool GEN>
__device__ __forceinline__ float unop(uint32_t sub, float x, float imm) {
    switch (sub) {
    case BSVI_U_COPY: return x;
    case BSVI_U_NEG: return -x;
    case BSVI_U_EXP: return expf(x);
    case BSVI_U_LOG: return logf(x);
    case BSVI_U_SQRT: return sqrtf(x);
    case BSVI_U_ABS: return fabsf(x);
    case BSVI_U_SIGMOID: return sigmoidf_(x);
    case BSVI_U_SOFTPLUS: return softplusf_(x);
    case BSVI_U_RELU: return fmaxf(x, 0.0f);
    case BSVI_U_RECIP: return 1.0f / x;
    case BSVI_U_SQUARE: return x * x;
    case BSVI_U_POWI:
        if (imm == 2.0f) return x * x;
        if (imm == -1.0f) return 1.0f / x;
        if (imm == 0.5f) return sqrtf(x);
        return GEN ? unop_rare(sub, x, imm) : x;
    default: return GEN ? unop_rare(sub, x, imm) : x;
    }
}
template <bool GEN>
__device__ __forceinline__ float unop_grad(uint32_t sub, float x, float y, float imm) {
    switch (sub) {
    case BSVI_U_COPY: return 1.0f;
    case BSVI_U_NEG: return -1.0f;
    case BSVI_U_EXP: return y;
    case BSVI_U_LOG: return 1.0f / x;
    case BSVI_U_SQRT: return 0.5f / y;
    case BSVI_U_ABS: return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    case BSVI_U_SIGMOID: return y * (1.0f - y);
    case BSVI_U_SOFTPLUS: return x > 20.0f ? 1.0f : sigmoidf_(x);
    case BSVI_U_RELU: return (x > 0.0f) ? 1.0f : 0.0f;
    case BSVI_U_RECIP: return -y * y;
    case BSVI_U_SQUARE: return 2.0f * x;
    case BSVI_U_POWI:
        if (imm == 2.0f) return 2.0f * x;
        if (imm == -1.0f) return -y * y;
        if (imm == 0.5f) return 0.5f / y;
        return GEN ? unop_rare_grad(sub, x, y, imm) : 1.0f;
    default: return GEN ? unop_rare_grad(sub, x, y, imm) : 1.0f;
    }
}

}  // namespace bsvi
