/*
 * bsvi.h — C ABI of the MI355X-native stochastic-variational-inference engine.
 *
 * This is the drop-in boundary for Brancher's ELBO-gradient hot path.  The reference
 * (LucaAmbrogioni/Brancher) is pure Python and has no FFI of its own; the seam this
 * library plugs into is the `InferenceMethod` interface of `brancher/inference.py:114-126`
 * (see INTEGRATION.md for the reference-side binding).  Each entry point below names the
 * reference code it replaces.
 *
 * Conventions
 *   - plain C types only; every `*_dev` pointer is a HIP device pointer owned by the caller
 *     (PyTorch-ROCm tensors in the Python host layer), `stream` is a hipStream_t passed as void*.
 *   - all functions return 0 on success or a negative bsvi_status; bsvi_last_error() gives
 *     a thread-local message.  Numerical failure (NaN/Inf loss) is NOT an error: it is
 *     reported through the finite flag of the output block so that the host can reproduce
 *     `inference.py:98,106-107` (warn and skip the optimizer step).
 *   - no function here synchronises the stream or allocates device memory except
 *     bsvi_program_create / bsvi_workspace_bytes users (allocation is the caller's job).
 *   - re-entrant per (program, workspace, stream); no global mutable state.
 *
 * Device data layout: structure-of-arrays with the Monte-Carlo sample axis
 * fastest — noise/eps, samples and per-sample outputs are [element][N] fp32, so the 64 lanes
 * of a wavefront (one lane = one MC sample) read consecutive addresses.
 */
#ifndef BSVI_H
#define BSVI_H

#if !defined(__HIPCC_RTC__)
#include <stddef.h>
#include <stdint.h>
#else   /* hiprtc (the library's run-time specialiser) has no system headers */
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long long uint64_t;
typedef long long int64_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define BSVI_ABI_VERSION 11

/* Every descriptor / argument struct a binder fills in starts with `struct_size` = sizeof of the struct AS THE BINDER
 * DECLARES IT.  An entry point that receives a struct whose struct_size differs from the library's own sizeof returns
 * BSVI_ERR_INVALID instead of reading past the end of a shorter struct (structs only ever grow at the tail, and a binding
 * written against an older header is otherwise indistinguishable from a current one).  bsvi_sizeof() answers the same
 * question ahead of time, also for the table element types that travel in arrays. */
typedef enum bsvi_struct_kind {
    BSVI_SK_UNIFORM_ENTRY = 0, BSVI_SK_RECORD = 1, BSVI_SK_PROGRAM_DESC = 2, BSVI_SK_ELBO_ARGS = 3, BSVI_SK_OPT_CFG = 4,
    BSVI_SK_DENSE_DESC = 5, BSVI_SK_DENSE_ARGS = 6, BSVI_SK_MLP_LAYER = 7, BSVI_SK_AMORT_DESC = 8, BSVI_SK_AMORT_ARGS = 9,
    BSVI_SK_MVN_INSN = 10, BSVI_SK_MVN_DESC = 11, BSVI_SK_MVN_ARGS = 12, BSVI_SK_BNN_LAYER = 13, BSVI_SK_BNN_DESC = 14,
    BSVI_SK_BNN_ARGS = 15, BSVI_SK_REDUCE_DESC = 16, BSVI_SK_REDUCE_ARGS = 17, BSVI_SK_COUNT = 18
} bsvi_struct_kind;
/* sizeof(struct) inside this build of the library; 0 for an unknown kind */
size_t bsvi_sizeof(int kind);

typedef enum bsvi_status {
    BSVI_OK = 0,
    BSVI_ERR_INVALID = -1,      /* malformed program / bad argument            */
    BSVI_ERR_UNSUPPORTED = -2,  /* program needs a feature this build lacks    */
    BSVI_ERR_HIP = -3,          /* a HIP runtime call failed                   */
    BSVI_ERR_NO_DEVICE = -4,    /* no gfx950 device visible                    */
    BSVI_ERR_RESOURCE = -5      /* program does not fit LDS / register budget  */
} bsvi_status;

/* ---- distributions: torch.distributions classes reached through
 *      brancher/distributions.py:476-592 (`self.torchdist`) ------------------------------ */
typedef enum bsvi_dist {
    BSVI_DIST_DETERMINISTIC = 0,
    BSVI_DIST_NORMAL = 1,
    BSVI_DIST_LOGNORMAL = 2,
    BSVI_DIST_CAUCHY = 3,
    BSVI_DIST_LAPLACE = 4,
    BSVI_DIST_BETA = 5,
    BSVI_DIST_BINOMIAL = 6,     /* p0 = total_count, p1 = logits */
    BSVI_DIST_BERNOULLI = 7,    /* p0 = logits                   */
    BSVI_DIST_CATEGORICAL = 8,
    /* not a distribution of the reference: the "log-density" p0 * x + p1, linear in its value.  Terms of this kind are how
     * a quantity computed OUTSIDE the per-sample program enters it with its value and its gradient: the batched
     * multivariate-normal kernel (bsvi_mvn_*) hands the program log p and d log p / d inputs of a sample as the rows of a
     * linear surrogate, and the program's reverse sweep carries the coefficients on to the latents and parameters. */
    BSVI_DIST_LINEAR = 9,
    BSVI_DIST_COUNT = 10
} bsvi_dist;

/* ---- the model program: a memory-to-memory instruction set over per-sample slots.
 *
 *  Every operand of every instruction is a memory operand, so the interpreter needs no
 *  register file (dynamic VGPR indexing spills on gfx950) and an instruction is fully
 *  described by one 32-byte slot fetched with a single scalar load:
 *
 *      w0 = opcode | flags<<8 | dist<<16 | rflags<<24     (rflags: BSVI_R_SINK | BSVI_R_NOALIAS)
 *      w1 = DST operand      (SAMPLE: the latent slot to write; LOGP: the VALUE operand)
 *      w2 = A operand   w3 = B operand   w4 = C operand   w5 = S operand
 *      w6 = imm0 (float bits)            w7 = imm1 (float bits)
 *
 *  operand word = byte_offset | walks<<30 | per_lane<<31      (resolved by the lowering)
 *      per_lane = 0  lane-uniform value in the uniform region of LDS: the uniform table U
 *                    (params / constants, transformed; entry k at byte 4k) followed by the
 *                    observed data (element i at byte 4*(n_uniform + i))
 *      per_lane = 1  a slot of this sample (latents, derived values, temps): slot s at byte 8s of
 *                    the sample's row — value at +0, adjoint at +4
 *      walks = 1     the operand advances with the element index e of the record's loop
 *                    (by 4 bytes for uniform entries / observed data, 8 bytes for slots)
 *      An absent factor / addend is encoded as the constant 1.0 / 0.0 of the uniform table.
 *
 *  BSVI_OP_NAFF   fused Normal node with affine location — the whole of
 *                 NormalVariable(loc = A*B + C, scale = S) in ONE instruction:
 *                 flags SAMPLE: z = loc + S*eps, DST slot <- z          (variables.py:527-570)
 *                       ENT:    f += imm1 * H[Normal(loc, S)]           (variables.py:156-162)
 *                       LOGP:   f += imm0 * log N(value | loc, S)       (variables.py:486-520)
 *                       WF:     lq += log N(value | loc, S)  (score term, gradient_estimators.py:33)
 *  BSVI_OP_NODE   the same for any other distribution `dist`, parameters p0 = A, p1 = B
 *  BSVI_OP_BIN    DST <- A (flags: add sub mul div pow delta) B         (variables.py:995-1002)
 *  BSVI_OP_UN     DST <- flags(A)  [imm0 = exponent of POWI]            (functions.py:28-41)
 *
 *  The program is ONE instruction stream.  A record (= one node evaluation) that is a single
 *  instruction over a single element is just that instruction; any other record is bracketed:
 *  BSVI_OP_REC_BEGIN / BSVI_OP_REC_END   w1 = body length, w2 = element-loop extent,
 *                 w3 = first temp slot, w4 = number of temp slots (their adjoints are re-zeroed
 *                 per element).  Both brackets carry the same words so that the reverse sweep
 *                 finds the record start from its end.
 *  BSVI_R_SINK    (rflags) the record is a model log-probability term: its weight is a constant
 *                 and nothing reads its value, so the kernel runs its forward AND reverse step
 *                 in the forward sweep (one visit, operands loaded once); the reverse sweep only
 *                 walks the posterior's sampling chain and the derived values.
 * --------------------------------------------------------------------------------------- */
typedef enum bsvi_op {
    BSVI_OP_NOP = 0, BSVI_OP_NAFF = 1, BSVI_OP_NODE = 2, BSVI_OP_BIN = 3, BSVI_OP_UN = 4,
    BSVI_OP_REC_BEGIN = 5, BSVI_OP_REC_END = 6
} bsvi_op;

#define BSVI_R_SINK 1u
#define BSVI_R_NOALIAS 2u   /* no two operands of the instruction share an adjoint cell */
/* (rflags bits 2..7 are reserved for the library: bsvi_program_create marks, in its device copy, the top-level
 *  NAFF instructions it pre-resolves into its internal address table — DESIGN.md §4.2) */

/* BSVI_F_GIVEN (with BSVI_F_SAMPLE): the node's noise row holds the VALUE of the variable, not its noise —
 * evaluation programs that score caller-supplied samples (importance weights, variables.py:821-841);
 * requires noise_dev. */
typedef enum bsvi_node_flags {
    BSVI_F_SAMPLE = 1, BSVI_F_ENT = 2, BSVI_F_LOGP = 4, BSVI_F_WF = 8, BSVI_F_GIVEN = 16
} bsvi_node_flags;

typedef enum bsvi_binop {
    BSVI_B_ADD = 0, BSVI_B_SUB = 1, BSVI_B_MUL = 2, BSVI_B_DIV = 3, BSVI_B_POW = 4,
    BSVI_B_DELTA = 5   /* (a == b) ? 1 : 0, brancher/utilities.py:357-358 */
} bsvi_binop;

typedef enum bsvi_unop {
    BSVI_U_COPY = 0, BSVI_U_NEG = 1, BSVI_U_EXP = 2, BSVI_U_LOG = 3, BSVI_U_SQRT = 4, BSVI_U_SIN = 5,
    BSVI_U_COS = 6, BSVI_U_TANH = 7, BSVI_U_ABS = 8, BSVI_U_SIGMOID = 9, BSVI_U_SOFTPLUS = 10,
    BSVI_U_RELU = 11, BSVI_U_RECIP = 12, BSVI_U_LOG1P = 13, BSVI_U_EXPM1 = 14, BSVI_U_SQUARE = 15,
    BSVI_U_P2L = 16,   /* probs -> logits with torch's clamp (torch/distributions/utils.py:127-137) */
    BSVI_U_POWI = 17   /* a ** imm0 */
} bsvi_unop;

/* uniform-table transforms: U[k] = a + b * g(src) (geometric_ranges.py forward transforms,
 * hoisted out of the per-sample program because they do not depend on the sample) */
typedef enum bsvi_utransform {
    BSVI_UT_IDENTITY = 0, BSVI_UT_SOFTPLUS = 1, BSVI_UT_SIGMOID = 2, BSVI_UT_EXP = 3,
    BSVI_UT_LOG = 4, BSVI_UT_TANH = 5, BSVI_UT_SQRT = 6, BSVI_UT_SQUARE = 7
} bsvi_utransform;

typedef struct bsvi_uniform_entry {
    uint32_t src;        /* index into params (is_param=1) or consts (is_param=0) */
    uint8_t  transform;  /* bsvi_utransform */
    uint8_t  is_param;
    uint16_t reserved;
    float    a, b;
} bsvi_uniform_entry;

/* one record = one node evaluation: the body span inside the instruction stream (the kernel
 * itself reads record boundaries from the stream; this table is what bsvi_program_create
 * validates the stream against) */
typedef struct bsvi_record {
    uint32_t code_begin, code_end;  /* body, in 32-byte instructions (brackets excluded) */
    uint32_t n_elems;               /* element loop extent (flattened B*D1*D2) */
    uint32_t temp_base, n_temps;    /* scratch slots this record writes (adjoints re-zeroed per element) */
    uint32_t flags;                 /* bit 0: BSVI_R_SINK */
} bsvi_record;

typedef enum bsvi_estimator {
    BSVI_EST_PATHWISE = 0,  /* gradient_estimators.py:39-44 */
    BSVI_EST_BLACKBOX = 1   /* gradient_estimators.py:29-36 */
    /* Taylor1Estimator (gradient_estimators.py:47-56) is BSVI_EST_PATHWISE on a different program: the host lowering
     * substitutes the posterior's analytic means for its values wherever the model reads them. */
} bsvi_estimator;

typedef struct bsvi_program_desc {
    uint32_t struct_size;     /* sizeof(bsvi_program_desc)                          */
    uint32_t abi_version;
    uint32_t n_params;        /* length of the flat parameter buffer                */
    uint32_t n_consts;        /* length of the constant buffer                      */
    uint32_t n_obs;           /* length of the observed-data buffer                 */
    uint32_t n_slots;         /* per-sample slots: latents, then derived, then temps */
    uint32_t n_noise;         /* latent rows = rows of the noise input / samples out */
    uint32_t n_uniform;       /* entries of the uniform table                       */
    uint32_t n_uniform_grad;  /* the first n_uniform_grad entries are param-sourced */
    uint32_t n_records;
    uint32_t n_code;          /* 32-byte instructions                               */
    uint32_t estimator;       /* bsvi_estimator                                     */
    const bsvi_uniform_entry* uniform;
    const bsvi_record* records;
    const uint32_t* code;     /* 8 * n_code words */
    const float* consts;      /* host copy; uploaded by bsvi_program_create */
    /* CSR map param -> uniform entries, for the deterministic chain rule U -> theta */
    const uint32_t* param_uniform_ptr;   /* n_params + 1 */
    const uint32_t* param_uniform_idx;   /* n_uniform_grad */
} bsvi_program_desc;

typedef struct bsvi_program bsvi_program;

/* Validate + upload an immutable program (replaces the per-call recursive graph walk of
 * brancher/variables.py:486-570,718-749 and the per-call name re-mapping of
 * brancher/utilities.py:282-309). */
int bsvi_program_create(const bsvi_program_desc* desc, bsvi_program** out);
void bsvi_program_destroy(bsvi_program* prog);

/* Bytes of device workspace bsvi_elbo_fwd_bwd needs for n_samples_local samples. */
size_t bsvi_workspace_bytes(const bsvi_program* prog, uint32_t n_samples_local);

/* Output block layout (fp32, length 4 + n_params):
 *   out[0] = sum over local samples of the per-sample estimator value (ELBO term, not yet /N)
 *   out[1] = number of non-finite per-sample values seen locally.  A diagnostic, not a count of samples: when the model's
 *            log-prob records are split over workgroups (program shares, bsvi_program_set_shares) every share tests its
 *            own partial value, so one bad sample can be counted once per share (<= n_shares times); out[3] is exact
 *   out[2] = loss = -out[0]/n_samples_global   (filled by bsvi_finalize)
 *   out[3] = finite flag (1 = finite)          (filled by bsvi_finalize)
 *   out[4..] = gradient sums d(sum f)/d theta  (bsvi_finalize scales them to d loss/d theta)
 */
#define BSVI_OUT_HEADER 4

typedef struct bsvi_elbo_args {
    uint32_t struct_size;         /* sizeof(bsvi_elbo_args)                                 */
    uint32_t reserved0;
    const float* params_dev;      /* [n_params]                                             */
    const float* obs_dev;         /* [n_obs]                                                */
    const float* noise_dev;       /* [n_noise][n_samples_local] or NULL -> in-kernel Philox  */
    uint64_t seed;                /* Philox key                                             */
    uint64_t offset;              /* Philox counter offset (iteration number)               */
    uint32_t n_samples_local;     /* samples evaluated by this call (this GPU's shard)      */
    uint32_t n_samples_global;    /* N of the estimator (all shards)                        */
    uint32_t sample_base;         /* global index of local sample 0 (Philox counter)        */
    uint32_t reserved;
    float* out_dev;               /* [BSVI_OUT_HEADER + n_params]                            */
    float* samples_out_dev;       /* [n_noise][n_samples_local] or NULL                     */
    float* noise_out_dev;         /* [n_noise][n_samples_local] or NULL (Philox draws used) */
    float* fvalue_out_dev;        /* [2][n_samples_local] or NULL: per-sample f and log q   */
    void* workspace_dev;          /* bsvi_workspace_bytes(...)                              */
    void* stream;
    const uint64_t* offset_dev;   /* NULL, or a device word added to `offset` when the launch executes: a step
                                     sequence captured in a HIP graph advances it on the device
                                     (bsvi_finalize_step_counted) and is replayed unchanged            */
    /* ABI 8 — caller-weighted gradients, the second pass of a user-defined GradientEstimator
     * (gradient_estimators.py:17-26: any scalar g(f, log q) of the per-sample values): with a_n = dg/df_n and
     * b_n = dg/dlog q_n the call leaves sum_n a_n grad f_n + b_n grad log q_n in out_dev[4..] (bsvi_elbo_fwd_bwd only;
     * the first pass reads f and log q through fvalue_out_dev, same seed and offset).  NULL: the estimator's own
     * weights (a_n = 1; BlackBox: b_n = f_n).  q_weight_dev needs a BlackBox program. */
    const float* f_weight_dev;    /* [n_samples_local] or NULL */
    const float* q_weight_dev;    /* [n_samples_local] or NULL */
} bsvi_elbo_args;

/* One ELBO forward+backward over this GPU's sample shard: q-sampling, p log-prob, q entropy
 * and the reverse sweep, fused (replaces ProbabilisticModel.estimate_log_model_evidence
 * brancher/variables.py:843-870, the estimators gradient_estimators.py:29-44 and
 * loss.backward() inference.py:100).  Leaves *sums* in out_dev[0,1,4..]. */
int bsvi_elbo_fwd_bwd(const bsvi_program* prog, const bsvi_elbo_args* args);

/* Turn the (all-reduced) sums into loss, finite flag and d loss/d theta
 * (the .mean() of gradient_estimators.py:36,44 and the sign of inference.py:141). */
int bsvi_finalize(const bsvi_program* prog, float* out_dev, uint32_t n_samples_global, void* stream);

/* ---- optimizer: torch.optim.{SGD,Adam} reached through brancher/optimizers.py:53-70 ---- */
typedef enum bsvi_optimizer_kind { BSVI_OPT_SGD = 0, BSVI_OPT_ADAM = 1 } bsvi_optimizer_kind;

typedef struct bsvi_opt_cfg {
    uint32_t kind;            /* bsvi_optimizer_kind */
    float lr;
    float momentum, dampening, weight_decay;   /* SGD */
    uint32_t nesterov;
    float beta1, beta2, eps;                    /* Adam */
    uint32_t amsgrad;
    uint32_t maximize;
} bsvi_opt_cfg;

/* Apply one optimizer step to params[i] for every i with active_mask[i] != 0, unless
 * out_dev[3] (finite flag) is 0 — `inference.py:98-107`.  `state_dev` is
 * [4][n_params] floats (momentum_buffer | exp_avg, exp_avg_sq, max_exp_avg_sq, step count —
 * torch keeps one step counter per parameter); zero-initialised by the caller. */
int bsvi_optimizer_step(const bsvi_opt_cfg* cfg, float* params_dev, const float* out_dev,
                        float* state_dev, const uint8_t* active_mask_dev, uint32_t n_params,
                        void* stream);

/* Multi-GPU step tail in ONE launch, after the caller has all-reduced the output block of bsvi_elbo_fwd_bwd
 * over the sample shards: out[2] = loss = -out[0]/n_samples_global, out[3] = finite flag, out[4..] scaled to
 * d loss/d theta, the optimizer step of every active parameter when the loss is finite
 * (brancher/inference.py:98-104), and the loss / finite flag stored to the (optional) slots
 * (inference.py:105-109).  Equivalent to bsvi_finalize + bsvi_optimizer_step + two device copies. */
int bsvi_finalize_step(const bsvi_opt_cfg* cfg, float* params_dev, float* out_dev, float* state_dev,
                       const uint8_t* active_mask_dev, uint32_t n_params, uint32_t n_samples_global,
                       float* loss_slot_dev, float* finite_slot_dev, void* stream);

/* bsvi_finalize_step for a step sequence captured ONCE in a HIP graph and replayed (the multi-GPU loop: fwd_bwd,
 * all-reduce, this): the iteration number lives on the device.  counters_dev[0] = Philox offset of the iteration (give
 * it to bsvi_elbo_fwd_bwd as bsvi_elbo_args::offset_dev), counters_dev[1] = index of the iteration within the run: it
 * selects loss_curve_dev[it] / finite_curve_dev[it] and the mask (active_mask_first_dev while it <=
 * pretraining_iterations, inference.py:102-104).  Both counters are advanced by the launch.  n_params <= 1024. */
int bsvi_finalize_step_counted(const bsvi_opt_cfg* cfg, float* params_dev, float* out_dev, float* state_dev,
                               const uint8_t* active_mask_dev, const uint8_t* active_mask_first_dev,
                               uint32_t pretraining_iterations, uint32_t n_params, uint32_t n_samples_global,
                               float* loss_curve_dev, float* finite_curve_dev, uint64_t* counters_dev, void* stream);

/* Run `n_iterations` complete SVI iterations (ELBO fwd+bwd, finalize, optimizer step, loss
 * log) inside one kernel launch when the local sample count fits one workgroup; the whole
 * loop of brancher/inference.py:95-108 without returning to the host.  loss_curve_dev gets
 * one loss per iteration, finite_dev one flag per iteration. */
int bsvi_train_persistent(const bsvi_program* prog, const bsvi_elbo_args* args,
                          const bsvi_opt_cfg* cfg, float* params_dev, float* state_dev,
                          const uint8_t* active_mask_dev, uint32_t n_iterations,
                          float* loss_curve_dev, float* finite_dev);

/* Same, with the reference's `pretraining_iterations` rule (inference.py:102-104): parameters
 * selected by active_mask_first_dev are stepped on every iteration, the rest of active_mask_dev
 * only when iteration > pretraining_iterations (the reference's second optimizer, built for the
 * joint model, is skipped on iteration 0 even with the default pretraining_iterations=0).
 * state_dev may be NULL when the call runs on the program-specialised kernel (bsvi_program_engine(prog, n, 2, ...) != 0):
 * the optimizer then starts from a fresh (all-zero) state kept inside the kernel and its final state is not returned —
 * what `perform_inference` needs, which builds its optimizers anew on every call (inference.py:77-88) — and the caller
 * neither allocates nor clears a state buffer.  On the interpreter kernels a NULL state_dev is BSVI_ERR_INVALID. */
int bsvi_train_persistent2(const bsvi_program* prog, const bsvi_elbo_args* args,
                           const bsvi_opt_cfg* cfg, float* params_dev, float* state_dev,
                           const uint8_t* active_mask_dev, const uint8_t* active_mask_first_dev,
                           uint32_t pretraining_iterations, uint32_t n_iterations,
                           float* loss_curve_dev, float* finite_dev);

/* The same loop on SEVERAL ranks (the Monte-Carlo samples of an iteration are sharded over the GPUs of a node):
 * every rank calls this with its shard (args->n_samples_local / sample_base; n_samples_global = the whole draw) and the
 * bsvi_exchange it shares with its peers (below), and runs n_iterations iterations in ONE launch — per iteration the
 * loss sum, the non-finite count and one gradient sum per parameter are exchanged INSIDE the kernel (the protocol of
 * bsvi_exchange_allreduce, run by the wave that owns the parameters: direct stores into every peer's region, sequence
 * numbers, slots added in rank order — bit-identical totals, so the replicated optimizer steps stay in lockstep), where
 * bsvi_elbo_fwd_bwd + bsvi_exchange_allreduce + bsvi_finalize_step_counted are three launches per iteration.  An
 * abandoned exchange (a peer did not arrive within the bound) leaves NaN in that iteration's loss, skips its optimizer
 * step on every rank and is sticky (bsvi_exchange_status).  Replaces the Python loop of inference.py:95-108 on N GPUs.
 * BSVI_ERR_UNSUPPORTED unless the shard runs on the program-specialised one-workgroup kernel (bsvi_program_engine, mode
 * 2), noise is Philox and no per-sample output is asked for: callers then fall back to the three-launch sequence.  (With at
 * most 64 parameters of at most two uniform-table entries each the wave that owns them exchanges; otherwise every thread
 * exchanges the entries of the parameters it steps.)
 * state_dev may be NULL (a fresh optimizer inside the kernel), as for bsvi_train_persistent2. */
struct bsvi_exchange;
int bsvi_train_persistent_exchange(const bsvi_program* prog, const bsvi_elbo_args* args, const bsvi_opt_cfg* cfg,
                                   float* params_dev, float* state_dev, const uint8_t* active_mask_dev,
                                   const uint8_t* active_mask_first_dev, uint32_t pretraining_iterations,
                                   uint32_t n_iterations, float* loss_curve_dev, float* finite_dev,
                                   struct bsvi_exchange* exchange);

/* 1 if bsvi_train_persistent supports (prog, n_samples_local), else 0. */
int bsvi_persistent_supported(const bsvi_program* prog, uint32_t n_samples_local);
/* The multi-workgroup persistent trainer (5..16 waves: one wave per workgroup, one exchange of partial sums per
 * iteration) can also split the MODEL's log-prob records over workgroups: every share samples the posterior but
 * evaluates only its part of the records; value and adjoints are linear in them, so the partial sums add up.
 * bsvi_persistent_split_shares: how many shares (1 = none, 2 or 3) the trainer would use for this sample count;
 * bsvi_train_persistent_split: shares[v] = programs created from the shares of the same lowering (identical tables,
 * own code), otherwise the arguments of bsvi_train_persistent2.  Pathwise estimator only. */
int bsvi_persistent_split_shares(const bsvi_program* p, uint32_t n_samples_local);
/* Attach up to 8 shares of `p` for the multi-workgroup launches of bsvi_elbo_fwd_bwd / bsvi_svi_step: when
 * (sample groups x shares) workgroups still get a CU each, workgroup b runs share b % n on sample group b / n and the
 * reduction adds the extra rows of partial sums.  n_shares < 2 detaches.  The shares must outlive their use. */
int bsvi_program_set_shares(bsvi_program* p, const bsvi_program* const* shares, uint32_t n_shares);
int bsvi_train_persistent_split(const bsvi_program* p, const bsvi_program* const* shares, uint32_t n_shares,
                                const bsvi_elbo_args* args, const bsvi_opt_cfg* cfg, float* params_dev, float* state_dev,
                                const uint8_t* active_mask_dev, const uint8_t* active_mask_first_dev,
                                uint32_t pretraining_iterations, uint32_t n_iterations, float* loss_curve_dev,
                                float* finite_dev);

/* One complete single-GPU SVI iteration in two launches: ELBO fwd+bwd, then reduction fused
 * with finalize and the optimizer step (inference.py:96-104).  Optionally logs the loss and the
 * finite flag of the iteration to loss_slot_dev / finite_slot_dev (inference.py:105). */
int bsvi_svi_step(const bsvi_program* prog, const bsvi_elbo_args* args, const bsvi_opt_cfg* cfg,
                  float* params_dev, float* state_dev, const uint8_t* active_mask_dev,
                  float* loss_slot_dev, float* finite_slot_dev);

/* ---- program specialisation -----------------------------------------------------------------------------------
 * bsvi_program_create also turns the instruction stream into straight-line HIP (every slot a register, every record
 * loop unrolled, weights and flags literals) inside a hand-written kernel frame; hiprtc compiles it for gfx950 at the
 * program's first launch, and bsvi_elbo_fwd_bwd / bsvi_svi_step / bsvi_train_persistent* run it in place of the
 * interpreter kernels (one launch per call: the reductions, the chain rule to theta, finalize and the optimizer step
 * are its epilogue).  The interpreter serves programs whose unrolled stream is too long, and everything when the
 * environment has BSVI_JIT=0.  Replaces the same reference code as bsvi_elbo_fwd_bwd.
 *
 * bsvi_program_source: the generated translation unit of a program (variant 0: training kernel, in-kernel Philox
 *   noise only; 1: diagnostic kernel with noise in / samples, noise and per-sample values out).  Host only — needs no
 *   device.  Returns the byte count including the terminator (0: not specialised, see bsvi_last_error) and copies
 *   when `capacity` suffices.
 * bsvi_jit_compile: compile such a translation unit for gfx950 (hiprtc; needs no device).
 * bsvi_jit_load: the code object of a translation unit the way a program's first launch obtains it — from this
 *   process's cache, else from the code-object cache on disk (keyed by source, embedded device headers, compile
 *   options, target and hiprtc / HIP runtime version; $BSVI_CACHE_DIR, else $XDG_CACHE_HOME/brancher_amd/jit, else
 *   ~/.cache/brancher_amd/jit; BSVI_JIT_CACHE=0 switches the disk cache off), else hiprtc, which then fills both.
 *   *origin: 1 hiprtc, 2 process cache, 3 disk cache.  Needs no device.
 * bsvi_jit_last_origin: the same code for the last specialised kernel this thread made ready (0: none yet).
 * bsvi_jit_cache_dir: the cache directory (byte count including the terminator; 1 = "", the disk cache is off).
 * bsvi_jit_compiler_identity: what the key holds about WHICH compiler will run — path, size and modification time of every
 *   loaded libamd_comgr / libhiprtc / libLLVM / libclang, in load order (a host that imported torch first compiles with the
 *   ROCm libraries bundled with torch, the same host under rocprofv3 with the system's: different code for the same source).
 * bsvi_program_engine: 1 when a call in `mode` (0 bsvi_elbo_fwd_bwd, 1 bsvi_svi_step, 2 bsvi_train_persistent*) over
 *   n_local samples is served by the specialised kernel (and its launch geometry), 0 when by the interpreter. */
size_t bsvi_program_source(const bsvi_program_desc* desc, int variant, char* buf, size_t capacity);
int bsvi_jit_compile(const char* source, size_t* code_bytes);
int bsvi_jit_load(const char* source, size_t* code_bytes, int* origin);
int bsvi_jit_last_origin(void);
size_t bsvi_jit_cache_dir(char* buf, size_t capacity);
size_t bsvi_jit_compiler_identity(char* buf, size_t capacity);
int bsvi_program_engine(const bsvi_program* prog, uint32_t n_local, int mode, uint32_t* n_blocks, uint32_t* n_threads,
                        uint32_t* lds_bytes);

/* LDS bytes per workgroup the runtime granted to this program's kernels (160 KiB on gfx950). */
int bsvi_max_lds_bytes(const bsvi_program* prog);

/* Launch geometry the library uses for n_samples_local samples (for tests/bench/DESIGN.md). */
int bsvi_query_geometry(const bsvi_program* prog, uint32_t n_samples_local, uint32_t* n_blocks,
                        uint32_t* n_waves, uint32_t* zglobal, uint64_t* lds_bytes);

/* =========================================================================================
 *  Dense-link path (BASELINE config 4): a mean-field Normal weight matrix W[C][P] whose use is
 *  the dense link  activations[n][b][c] = sum_p W[n][c][p] * x[b][p]  (`BF.matmul(weights, x)`,
 *  examples/MNIST_logistic_regression.py:33; brancher/functions.py:28-41) feeding an observed
 *  likelihood, with the minibatch x drawn per iteration by RandomIndices / EmpiricalVariable
 *  (standard_variables.py:71-112, distributions.py:410-462).  The per-sample weights
 *  W = loc + scale * eps are never materialised: the forward GEMM [N*C x P] x [P x B] builds its
 *  A operand from eps on the fly and runs on the f32 matrix cores (v_mfma_f32_16x16x4_f32), the
 *  likelihood (log-softmax cross-entropy / Bernoulli logits) is its epilogue, and the backward
 *  is a second MFMA GEMM  T_c = eps_c^T x dlogits_c  reduced against x.
 * ========================================================================================= */
typedef enum bsvi_dense_likelihood {
    BSVI_LIK_CATEGORICAL = 0,   /* CategoricalVariable(logits=...)            distributions.py:294-311 */
    BSVI_LIK_BERNOULLI = 1      /* Binomial(1, logits=...) / Bernulli(logits) distributions.py:561-592 */
} bsvi_dense_likelihood;

typedef struct bsvi_dense_desc {
    uint32_t struct_size;                      /* sizeof(bsvi_dense_desc) */
    uint32_t abi_version;
    uint32_t n_params, n_consts, n_uniform, n_uniform_grad;
    uint32_t n_classes, n_features, dataset_size, batch_size;
    uint32_t likelihood;                       /* bsvi_dense_likelihood */
    /* rows r = c * n_features + p of the weight matrix read their parameters from the uniform
     * table at  base + r * stride  (stride 0 = one scalar for all rows) */
    uint32_t q_loc_u, q_scale_u, prior_loc_u, prior_scale_u;
    uint32_t q_loc_stride, q_scale_stride, prior_loc_stride, prior_scale_stride;
    float lik_weight, prior_weight, entropy_weight;
    uint32_t estimator;                        /* bsvi_estimator: pathwise, or BlackBox (gradient_estimators.py:29-36) */
    uint32_t reserved;
    const bsvi_uniform_entry* uniform;
    const float* consts;
    const uint32_t* param_uniform_ptr;
    const uint32_t* param_uniform_idx;
    const float* dataset;                      /* [dataset_size][n_features] host copy */
    const float* labels;                       /* [dataset_size] host copy            */
} bsvi_dense_desc;

typedef struct bsvi_dense bsvi_dense;

typedef struct bsvi_dense_args {
    uint32_t struct_size;        /* sizeof(bsvi_dense_args)                                      */
    uint32_t reserved0;
    const float* params_dev;     /* [n_params]                                                   */
    const float* noise_dev;      /* eps [C*P][n_samples_local] or NULL -> Philox                  */
    const int32_t* indices_dev;  /* minibatch rows [batch_size] or NULL -> drawn on the device    */
    uint64_t seed, offset;
    uint32_t n_samples_local, n_samples_global, sample_base, reserved;
    float* out_dev;              /* [BSVI_OUT_HEADER + n_params], same layout as bsvi_elbo_fwd_bwd */
    float* noise_out_dev;        /* eps used [C*P][n_samples_local] or NULL                       */
    int32_t* indices_out_dev;    /* minibatch used [batch_size] or NULL                           */
    float* fvalue_out_dev;       /* per-sample f [n_samples_local] or NULL                        */
    void* workspace_dev;
    void* stream;
    /* Caller-weighted gradients, as bsvi_elbo_args::f_weight_dev / q_weight_dev (the two passes of a user-defined
     * GradientEstimator, gradient_estimators.py:17-26): with both given the output block receives
     * sum_n a_n grad f_n + b_n grad log q_n of the same draw (seed, offset).  NULL: the estimator's own weights
     * (a_n = 1; BlackBox: b_n = f_n).  q_weight_dev and logq_out_dev need a model created with the BlackBox
     * estimator, which takes the two weights together or not at all.  Not accepted by bsvi_dense_step. */
    const float* f_weight_dev;   /* [n_samples_local] or NULL                                     */
    const float* q_weight_dev;   /* [n_samples_local] or NULL                                     */
    float* logq_out_dev;         /* per-sample log q(W_n) [n_samples_local] or NULL               */
} bsvi_dense_args;

int bsvi_dense_create(const bsvi_dense_desc* desc, bsvi_dense** out);
void bsvi_dense_destroy(bsvi_dense* d);
size_t bsvi_dense_workspace_bytes(const bsvi_dense* d, uint32_t n_samples_local);
/* 1 when every dataset value is exactly a bf16 number (pixel counts 0..255, binarised images) and the features come in whole
 * quads: both products of the iteration then have one exact operand — the minibatch — and run on the bf16 matrix cores as
 * three MFMAs on the exact pieces hi + mid + lo of the other operand (W = mu + s * eps, d f / d logits), products exact,
 * f32 accumulation; the noise is never stored (both consumers draw it from the same Philox counters).  0: the f32-input MFMA
 * kernels serve the model (also with BSVI_DENSE_XGEMM=0 at create time). */
int bsvi_dense_exact_data(const bsvi_dense* d);
/* ELBO forward+backward of the dense model over this GPU's sample shard; leaves sums in out_dev
 * (then bsvi_finalize / all-reduce / bsvi_optimizer_step as for bsvi_elbo_fwd_bwd). */
int bsvi_dense_fwd_bwd(const bsvi_dense* d, const bsvi_dense_args* args);
int bsvi_dense_finalize(const bsvi_dense* d, float* out_dev, uint32_t n_samples_global, void* stream);
/* single-GPU iteration with the reduction fused with finalize + optimizer step */
int bsvi_dense_step(const bsvi_dense* d, const bsvi_dense_args* args, const bsvi_opt_cfg* cfg,
                    float* params_dev, float* state_dev, const uint8_t* active_mask_dev,
                    float* loss_slot_dev, float* finite_slot_dev);

/* =========================================================================================
 *  Bayesian neural networks on the dense-link path (the reference's tests/test_MNIST_bayesian_neural_network.py:20-60):
 *      logits = W_L act( ... act(W_1 x + b_1) ... ) + b_L        (`BF.tanh(BF.matmul(weights1, x) + b1)`, functions.py:28-41)
 *  with EVERY weight matrix and bias a latent under a mean-field Normal posterior, an observed Categorical / Binomial(1)
 *  likelihood and a random minibatch per iteration.  All latent scalars form one vector of n_rows rows — weights1 first
 *  (H_1 x P, row r = h * P + p), the other tensors behind it — whose noise is the dense path's [n_rows][N] matrix (Philox
 *  counters (sample, row >> 2)); every row reads its q loc / q scale / prior loc / prior scale from the uniform table through
 *  row_uniform.  The two products with the minibatch (2 x 2 N H_1 P B flops) run on the matrix cores — on exact bf16 pieces when
 *  the dataset is exactly bf16 (bsvi_bnn_exact_data), else on the f32-input MFMA kernels; the upper layers are per-sample
 *  H_l x H_l-1 products evaluated per (sample, minibatch row).  Replaces the same reference lines as bsvi_dense_* for this graph.
 * ========================================================================================= */
typedef enum bsvi_bnn_activation {
    BSVI_BNN_ACT_NONE = 0, BSVI_BNN_ACT_TANH = 1, BSVI_BNN_ACT_RELU = 2, BSVI_BNN_ACT_SIGMOID = 3, BSVI_BNN_ACT_SOFTPLUS = 4
} bsvi_bnn_activation;

typedef struct bsvi_bnn_layer {
    uint32_t rows, cols;             /* W_l is [rows][cols]; cols = rows of the layer below (n_features for the first) */
    uint32_t weight_row0;            /* first row of W_l in the latent vector, row-major (0 for the first layer)         */
    uint32_t bias_row0;              /* first row of b_l, or 0xFFFFFFFF: no bias                                          */
    uint32_t activation;             /* bsvi_bnn_activation of the layer's output; NONE for the last (the logits)         */
    uint32_t reserved;
} bsvi_bnn_layer;

typedef struct bsvi_bnn_desc {
    uint32_t struct_size;            /* sizeof(bsvi_bnn_desc) */
    uint32_t abi_version;
    uint32_t n_params, n_consts, n_uniform, n_uniform_grad;
    uint32_t n_layers, n_rows, n_features, dataset_size, batch_size;
    uint32_t likelihood;             /* bsvi_dense_likelihood */
    uint32_t estimator;              /* bsvi_estimator: pathwise or BlackBox */
    float lik_weight, prior_weight, entropy_weight;
    const bsvi_bnn_layer* layers;    /* [n_layers], bottom up */
    const uint32_t* row_uniform;     /* [4][n_rows]: uniform entry of q loc, q scale, prior loc, prior scale of every row */
    const bsvi_uniform_entry* uniform;
    const float* consts;
    const uint32_t* param_uniform_ptr;
    const uint32_t* param_uniform_idx;
    const float* dataset;            /* [dataset_size][n_features] host copy */
    const float* labels;             /* [dataset_size] host copy            */
} bsvi_bnn_desc;

typedef struct bsvi_bnn bsvi_bnn;

typedef struct bsvi_bnn_args {
    uint32_t struct_size;            /* sizeof(bsvi_bnn_args) */
    uint32_t reserved0;
    const float* params_dev;         /* [n_params]                                                   */
    const float* noise_dev;          /* eps [n_rows][n_samples_local] or NULL -> Philox               */
    const int32_t* indices_dev;      /* minibatch rows [batch_size] or NULL -> drawn on the device    */
    uint64_t seed, offset;
    uint32_t n_samples_local, n_samples_global, sample_base, reserved;
    float* out_dev;                  /* [BSVI_OUT_HEADER + n_params], same layout as bsvi_elbo_fwd_bwd */
    float* noise_out_dev;            /* eps used [n_rows][n_samples_local] or NULL                    */
    int32_t* indices_out_dev;        /* minibatch used [batch_size] or NULL                           */
    float* fvalue_out_dev;           /* per-sample f [n_samples_local] or NULL                        */
    float* logq_out_dev;             /* per-sample log q [n_samples_local] or NULL (BlackBox)         */
    void* workspace_dev;
    void* stream;
    /* ABI 11 — caller-weighted gradients, as bsvi_elbo_args / bsvi_dense_args::f_weight_dev, q_weight_dev (the two passes of a
     * user-defined GradientEstimator, gradient_estimators.py:17-26): the call leaves sum_n a_n grad f_n + b_n grad log q_n in
     * out_dev[4..].  NULL: the estimator's own weights (a_n = 1; BlackBox: b_n = f_n).  q_weight_dev needs a model created with the
     * BlackBox estimator, which takes both or neither. */
    const float* f_weight_dev;       /* [n_samples_local] or NULL */
    const float* q_weight_dev;       /* [n_samples_local] or NULL */
} bsvi_bnn_args;

int bsvi_bnn_create(const bsvi_bnn_desc* desc, bsvi_bnn** out);
void bsvi_bnn_destroy(bsvi_bnn* b);
size_t bsvi_bnn_workspace_bytes(const bsvi_bnn* b, uint32_t n_samples_local);
int bsvi_bnn_exact_data(const bsvi_bnn* b);
/* ELBO forward + backward over this GPU's sample shard; sums in out_dev (then all-reduce / bsvi_bnn_finalize / bsvi_finalize_step) */
int bsvi_bnn_fwd_bwd(const bsvi_bnn* b, const bsvi_bnn_args* args);
int bsvi_bnn_finalize(const bsvi_bnn* b, float* out_dev, uint32_t n_samples_global, void* stream);
/* single-GPU iteration: the same launches with finalize + optimizer step fused into the last one */
int bsvi_bnn_step(const bsvi_bnn* b, const bsvi_bnn_args* args, const bsvi_opt_cfg* cfg, float* params_dev, float* state_dev,
                  const uint8_t* active_mask_dev, float* loss_slot_dev, float* finite_slot_dev);

/* =========================================================================================
 *  Amortised path (BASELINE config 5, examples/VAE_playground.py:18-88): the posterior of a latent
 *  vector z is a Normal whose loc / scale are the heads of an encoder network applied to a minibatch
 *  row x that EVERY Monte-Carlo sample draws for itself (EmpiricalVariable(dataset, batch_size=B),
 *  distributions.py:410-457 — number_samples * B rows per iteration), and the likelihood of x is
 *  Binomial(1, logits = decoder(z)).  Both networks are `BF.BrancherFunction(torch.nn.Module)` links
 *  (functions.py:15-41) made of Linear layers and elementwise activations; their weights are segments
 *  of the flat parameter buffer in torch's own layout (Linear.weight [n_out][n_in], bias [n_out]).
 *  Every Linear layer is an f32 MFMA GEMM over all R = n_samples * B rows (forward x W^T, backward-data
 *  dY W and backward-weight dY^T x), with bias + activation (forward) and the activation derivative
 *  (backward) as GEMM epilogues; the first layer gathers its rows from the dataset through the
 *  minibatch indices, so the minibatch is never materialised.
 *  Per row:  f = log p(x | z) + log p(z) + H[q(z | x)]  -> [N, B]; ELBO estimate = mean over N*B
 *  (variables.py:851-855, gradient_estimators.py:29-44).  estimator 0 = pathwise, 1 = BlackBox.
 * ========================================================================================= */
typedef enum bsvi_mlp_activation {
    BSVI_ACT_NONE = 0,
    BSVI_ACT_RELU = 1,
    BSVI_ACT_SOFTPLUS = 2
} bsvi_mlp_activation;

/* y[out_value] = activation(x[in_value] W^T + b) + post_add;  value 0 is the network's input */
typedef struct bsvi_mlp_layer {
    uint32_t in_value, out_value;
    uint32_t n_in, n_out;
    uint32_t weight_off, bias_off;   /* offsets into the flat parameter buffer; bias_off = 0xFFFFFFFF: no bias */
    uint32_t activation;             /* bsvi_mlp_activation */
    float post_add;
    /* sibling layers reading the same value (the latent's loc and scale heads) may be given as ONE layer whose weight
     * rows are adjacent in the parameter buffer: output columns >= split_col use activation2 / post_add2.
     * split_col >= n_out (or 0): no split. */
    uint32_t split_col;
    uint32_t activation2;
    float post_add2;
    uint32_t reserved;
} bsvi_mlp_layer;

typedef struct bsvi_amort_desc {
    uint32_t struct_size, reserved0;              /* sizeof(bsvi_amort_desc) */
    uint32_t abi_version, n_params;
    uint32_t n_features, latent_dim, dataset_size, batch_size;
    uint32_t n_enc_layers, n_dec_layers;          /* topologically ordered */
    uint32_t enc_loc_value, enc_scale_value;      /* encoder values feeding q(z | x) = Normal(loc, scale)  */
    uint32_t enc_loc_col, enc_scale_col;          /* first column of loc / scale inside those values (merged heads) */
    uint32_t dec_logits_value, likelihood;        /* decoder value feeding the likelihood; BSVI_AMORT_LIK_*  */
    const bsvi_mlp_layer* enc_layers;
    const bsvi_mlp_layer* dec_layers;
    const float* prior_loc;                       /* [latent_dim] host: p(z) = Normal(prior_loc, prior_scale) */
    const float* prior_scale;
    const float* dataset;                         /* [dataset_size][n_features] host copy */
    /* ABI 7 */
    const float* likelihood_scale;                /* [n_features] host, BSVI_AMORT_LIK_NORMAL: x ~ Normal(decoder value, scale) */
    /* a learnable prior (`NormalVariable(..., learnable=True)`, standard_variables.py:57-68): offsets of its raw values in
     * the parameter buffer — loc as stored, scale = softplus(raw) (geometric_ranges.py RightHalfLine) — or
     * BSVI_AMORT_CONSTANT: the constants above */
    uint32_t prior_loc_off, prior_scale_off;
    /* ABI 10: a learnable scale of the Normal likelihood (`NormalVariable(decoder value, scale, learnable=True)`): offset of
     * its raw values in the parameter buffer — scale = softplus(raw) — and their count (1: one scale for every feature, or
     * n_features); BSVI_AMORT_CONSTANT: the constants of likelihood_scale (which still carries the initial values) */
    uint32_t lik_scale_off, lik_scale_size;
    /* ... or a second HEAD of the decoder (`NormalVariable(decoder(z)["mean"], decoder(z)["sd"])`): the decoder value that holds
     * the scale (width n_features, a leaf, positive through its layer's activation), 0: none.  The row kernel leaves
     * d log p / d (pre-activation) in that value's gradient buffer, the decoder's backward pass takes it from there. */
    uint32_t dec_scale_value, reserved1;
} bsvi_amort_desc;
#define BSVI_AMORT_LIK_BINOMIAL1 0u               /* Binomial(1, logits = decoder value)  (examples/VAE_playground.py:71) */
#define BSVI_AMORT_LIK_NORMAL 1u
#define BSVI_AMORT_CONSTANT 0xFFFFFFFFu

typedef struct bsvi_amort bsvi_amort;

typedef struct bsvi_amort_args {
    uint32_t struct_size;         /* sizeof(bsvi_amort_args)                                            */
    uint32_t reserved0;
    const float* params_dev;      /* [n_params]                                                         */
    const float* noise_dev;       /* eps [n_samples_local * B][latent_dim] or NULL -> Philox             */
    const int32_t* indices_dev;   /* minibatch rows [n_samples_local][B] or NULL -> drawn on the device  */
    uint64_t seed, offset;
    uint32_t n_samples_local, n_samples_global, sample_base, estimator;
    float* out_dev;               /* [BSVI_OUT_HEADER + n_params]: out[0] = sum over rows of the estimator value,
                                     out[1] = non-finite rows, out[4..] = gradient sums; bsvi_finalize_step with
                                     n_samples_global * batch_size turns the sums into loss and gradients        */
    float* noise_out_dev;         /* eps used, or NULL                    */
    int32_t* indices_out_dev;     /* minibatch used, or NULL              */
    float* fvalue_out_dev;        /* f per row [n_samples_local * B], or NULL */
    float* logq_out_dev;          /* log q(z | x) per row, or NULL        */
    void* workspace_dev;
    void* stream;
    /* Caller-weighted gradients (the two passes of a user-defined GradientEstimator, gradient_estimators.py:17-26, as
     * bsvi_elbo_args::f_weight_dev / q_weight_dev): with both given — one value per ROW, [n_samples_local * B] — the
     * output block receives sum_r a_r grad f_r + b_r grad log q_r of the same draw and minibatches (seed, offset);
     * `estimator` then only selects the value summed into out[0].  Both or neither. */
    const float* f_weight_dev;
    const float* q_weight_dev;
} bsvi_amort_args;

int bsvi_amort_create(const bsvi_amort_desc* desc, bsvi_amort** out);
void bsvi_amort_destroy(bsvi_amort* a);
/* Workspace of one bsvi_amort_fwd_bwd call over n_samples_local samples: per-row values and gradients of every layer, and
 * behind them the per-slice partials of every reduction over rows (weight / bias gradients, loss sums), which one launch at
 * the end of the call adds in slice order — the output block is bit-reproducible call to call, no float atomics. */
size_t bsvi_amort_workspace_bytes(const bsvi_amort* a, uint32_t n_samples_local);
/* 1 when every value of the dataset is exactly a bf16 number (binarised images as in examples/VAE_playground.py:22-26,
 * pixel counts): the layers that read the data rows then run on the bf16 matrix cores — x W^T as three bf16 MFMAs on the
 * exact pieces hi + mid + lo of the f32 weights, products exact, f32 accumulation — instead of the f32-input MFMA;
 * 0: the f32 kernels serve them (also with BSVI_AMORT_XGEMM=0 at create time). */
int bsvi_amort_exact_data(const bsvi_amort* a);
int bsvi_amort_fwd_bwd(const bsvi_amort* a, const bsvi_amort_args* args);

/* Forward pass of one network (0 = encoder, 1 = decoder) on caller-supplied rows input_dev [n_rows][input width]; the
 * network value `value` (a layer output id) is written to out_dev [n_rows][its width].  Posterior predictive / encoding:
 * examples/VAE_playground.py:90-103.  workspace_dev: bsvi_amort_workspace_bytes(a, ceil(n_rows / batch_size)) bytes. */
int bsvi_amort_apply(const bsvi_amort* a, int network, const float* params_dev, const float* input_dev, uint32_t n_rows,
                     uint32_t value, float* out_dev, void* workspace_dev, void* stream);

/* Several ranks: the gradients of the DECODER complete early.  The backward pass of bsvi_amort_fwd_bwd finishes the decoder's
 * weight gradients before it starts on the encoder's.  bsvi_amort_bucket reports the decoder's parameters (and the likelihood's
 * scale) as one range [first_param, first_param + n_params) of the parameter vector (n_params = 0: they are not one range).  With a
 * bucket stream set, bsvi_amort_fwd_bwd reduces that range's partial sums on THAT stream as soon as the kernels producing them are
 * in flight (the stream waits for them through events), so that the host's all-reduce of out[4 + first_param ...] — enqueued on the
 * bucket stream right after the call — runs beside the encoder's backward pass; everything else is reduced at the end of the call on
 * args->stream as always.  The caller joins the two streams before bsvi_finalize_step.  NULL (the default): one reduction at the end.
 * What torch.distributed's bucketed all-reduce does for the reference's PyTorch modules under DistributedDataParallel. */
int bsvi_amort_bucket(const bsvi_amort* a, uint32_t* first_param, uint32_t* n_params);
int bsvi_amort_set_bucket_stream(bsvi_amort* a, void* stream);

/* Test hook: one launch of the f32 MFMA GEMM behind the amortised path.
 * mode 0: C[M][N] = A[M][K] B[N][K]^T   (+ bias[n], activation)        forward
 * mode 1: C[M][N] = A[M][K] B[K][N]     (* activation'(Y[m][n]))      backward-data
 * mode 2: C[M][N] += A[K][M]^T B[K][N]  (K split over workgroups: per-slice partials in a buffer the hook owns, added
 *         in slice order by a second launch) backward-weight
 * modes 5 / 6: modes 0 / 1 as six products of exact bf16 pieces on the bf16 matrix cores (the wide layers' product path
 *         from 256 rows; K and lda multiples of 4, no gather)
 * rows_dev (or NULL) gathers the rows of A (modes 0, 1) / of B (mode 2).
 * bias_or_y_dev: mode 0 bias[N]; mode 1 Y[M][ldy]; mode 2 an [M] accumulator that receives += column sums of A. */
int bsvi_debug_gemm(int mode, const float* a_dev, const float* b_dev, float* c_dev, const int32_t* rows_dev,
                    uint32_t m, uint32_t n, uint32_t k, uint32_t lda, uint32_t ldb, uint32_t ldc,
                    const float* bias_or_y_dev, uint32_t ldy, uint32_t activation, float post_add,
                    uint32_t accumulate, void* stream);

/* Test hook, not used by the product path: evaluates one special function (fn 0 digamma,
 * 1 trigamma, 2 dirichlet_grad_one(x, alpha=p0, total=p1), 6 lgamma) or one node function of
 * distribution `dist` (fn 3 log-prob, 4 entropy, 5 reparameterised draw from noise x) elementwise;
 * out_dev is [4][n] = value, d/dx, d/dp0, d/dp1. */
int bsvi_debug_math(int fn, int dist, const float* x_dev, const float* p0_dev, const float* p1_dev,
                    float* out_dev, uint32_t n, void* stream);

/* Diagnostic hook (tools/phase_stamps.py): 10 uint64 = (s_memtime, s_memrealtime) at the phase
 * boundaries prologue / forward / backward / reduction of workgroup 0; NULL switches it off. */
void bsvi_debug_set_stamps(unsigned long long* stamps_dev);

/* =========================================================================================
 * Batched multivariate-normal nodes (brancher/distributions.py:314-331,
 * brancher/standard_variables.py:317-347 with `covariance_matrix`).
 *
 * log N(x | m, C) and its gradient for a covariance that is an ELEMENTWISE link expression of constant matrices and a
 * few scalars that differ per Monte-Carlo sample (latent kernel hyper-parameters) or are learnable — the case the
 * per-sample program cannot hold (D^2 values per sample).  One workgroup of four waves per sample factorises C (Cholesky,
 * triangular inverse, C^-1 = L^-T L^-1) and contracts d log p / d C = (alpha alpha^T - C^-1) / 2 with the expression's
 * derivatives.  The covariance expression arrives as three-address code (temp t = instruction t; the last instruction
 * is C_ij); the library generates HIP from it and compiles it with hiprtc at the first evaluation (same caches as the
 * program-specialised ELBO kernels).  dim <= 1024: up to 192 the one matrix of a sample (L, its inverse and C^-1 share it) is
 * in LDS; beyond that it is a block of device memory the node allocates at its first evaluation (padded dim^2 floats per
 * sample of the largest launch so far; freed by bsvi_mvn_destroy — one block per node, so evaluations of ONE node with dim > 192
 * must be ordered on a stream, not run on two streams at once).  At most 8 scalar inputs.
 *
 * The node talks to the per-sample program through rows of per-sample values:
 *   in   samples_dev [rows][n_local]: the slot values of the draw (samples_out of a bsvi_elbo_fwd_bwd call);
 *        input_rows[k] is the row of scalar input k < n_slot_inputs, value_row0 the first of the dim rows of x when x is
 *        latent (value_is_latent), params_dev feeds the uniform inputs (a + b * g(params[src]))
 *   out  rows_out_dev [bsvi_mvn_rows_out][n_local]: g_k = weight * d log p / d input_k for the slot inputs, then for the
 *        dim elements of a latent x, then for the uniform inputs, then for the dim elements of a learnable loc, then
 *        e = weight * log p - sum_k g_k * input_k.
 *        e + sum_k g_k * input_k has the value and the gradient of weight * log p at the sample: the program adds it to f
 *        with BSVI_DIST_LINEAR terms whose coefficients are BSVI_F_GIVEN rows.  A covariance that is not positive
 *        definite gives NaN rows (the step is then skipped as non-finite, brancher/inference.py:98). */
typedef enum bsvi_mvn_kind { BSVI_MVN_MAT = 0, BSVI_MVN_INPUT = 1, BSVI_MVN_IMM = 2, BSVI_MVN_BIN = 3, BSVI_MVN_UN = 4 } bsvi_mvn_kind;
typedef struct bsvi_mvn_insn {
    uint32_t kind;     /* bsvi_mvn_kind */
    uint32_t flag;     /* BIN: bsvi_binop, UN: bsvi_unop */
    uint32_t a, b;     /* MAT: a = matrix index; INPUT: a = input index (slot inputs first, then uniform inputs); BIN / UN: temps */
    float imm;         /* IMM: the value; UN POWI: the exponent */
} bsvi_mvn_insn;
typedef struct bsvi_mvn_desc {
    uint32_t struct_size, reserved0;           /* sizeof(bsvi_mvn_desc) */
    uint32_t abi_version, dim, n_code, n_mats;
    uint32_t n_slot_inputs, n_uniform_inputs, value_is_latent, loc_is_param;
    const bsvi_mvn_insn* code;                 /* host */
    const float* mats;                         /* host [n_mats][dim][dim] */
    const float* loc;                          /* host [dim] */
    const float* value;                        /* host [dim]: the observed x (value_is_latent == 0) */
    const bsvi_uniform_entry* uniform_inputs;  /* host [n_uniform_inputs], parameter-sourced */
    const bsvi_uniform_entry* loc_entries;     /* host [dim]: a LEARNABLE loc (loc_is_param; by the reference's name-collision rule
                                                  the prior's loc root is often the posterior's learnable mean) */
    float weight;                              /* of log p in f */
    uint32_t form;                             /* bsvi_mvn_form: what the expression yields (distributions.py:314-331) */
    /* ABI 10: a value that is itself [dim] transformed PARAMETERS (value_is_latent == 0, value_is_param != 0) — the taylor1
     * program reads the model at the posterior's means (gradient_estimators.py:47-56), and the mean of Normal(loc, scale) is
     * its learnable loc; `value` may then be NULL.  Its coefficient rows -alpha stand where a latent value's would. */
    const bsvi_uniform_entry* value_entries;   /* host [dim] */
    uint32_t value_is_param, reserved1;
} bsvi_mvn_desc;
typedef enum bsvi_mvn_form {
    BSVI_MVN_COVARIANCE = 0,    /* MultivariateNormalVariable(covariance_matrix=...): factorised per sample              */
    BSVI_MVN_SCALE_TRIL = 1,    /* (scale_tril=...): the expression IS the Cholesky factor (lower triangle): no factorisation */
    BSVI_MVN_PRECISION = 2      /* (precision_matrix=...): the precision is factorised; its inverse is the covariance    */
} bsvi_mvn_form;
typedef struct bsvi_mvn_args {
    uint32_t struct_size, reserved0;           /* sizeof(bsvi_mvn_args) */
    const float* params_dev;
    const float* samples_dev;
    float* rows_out_dev;
    uint32_t n_samples_local, value_row0;
    uint32_t input_rows[8];
    void* stream;
} bsvi_mvn_args;
typedef struct bsvi_mvn bsvi_mvn;
int bsvi_mvn_create(const bsvi_mvn_desc* desc, bsvi_mvn** out);
void bsvi_mvn_destroy(bsvi_mvn* m);
uint32_t bsvi_mvn_rows_out(const bsvi_mvn_desc* desc);
int bsvi_mvn_eval(bsvi_mvn* m, const bsvi_mvn_args* args);
/* the generated translation unit (host only, like bsvi_program_source): byte count including the terminator, 0 on error */
size_t bsvi_mvn_source(const bsvi_mvn_desc* desc, char* buf, size_t capacity);

/* =========================================================================================
 * The REDUCE node (ABI 11): links with a reduction over more elements than the per-sample program unrolls — the last of the
 * reference's examples that did not lower, examples/PopulationReceptiveFields.py:29-31:
 *     mean_response = BF.sum(BF.sum(receptive_field * input, dim=1, keepdim=True), dim=2, keepdim=True)
 * i.e. r[n][d] = sum_e A(e; s_n) * X[d][e] with A an ELEMENTWISE link expression of constant matrices (rows x cols each) and at most
 * 8 scalars that differ per Monte-Carlo sample or are learnable (three-address code, as the multivariate-normal node's), and X a data
 * matrix [n_data][rows * cols]: a constant of the model (drawn == 0: data_mean IS X), or — drawn != 0 — the value of a Normal node that
 * is observed BY FLAG only, which the reference draws from Normal(data_mean, data_scale) ONCE per evaluation for all samples
 * (brancher/variables.py:849 takes observed_submodel._get_sample(1, observed=True); :553-565 draws what has no value): drawn on the
 * device from (seed, offset), or handed in (bsvi_reduce_args::data_dev: parity tests replay the reference's draw).
 *
 * Rows out, [bsvi_reduce_rows_out][n_local]: for input k and datapoint d, row k * n_data + d = g_dk = d r_d / d input_k; then row
 * n_inputs * n_data + d = e_d = r_d - sum_k g_dk input_k; then ONE row: weight * log N(X | data_mean, data_scale) (drawn; else 0).
 * e_d + sum_k g_dk input_k has the value and the gradient of r_d at the sample: the per-sample program composes r_d from these GIVEN
 * rows (lowering.reduce_external) wherever the link stood. */
typedef struct bsvi_reduce_desc {
    uint32_t struct_size, reserved0;           /* sizeof(bsvi_reduce_desc) */
    uint32_t abi_version, rows, cols, n_code, n_mats;
    uint32_t n_slot_inputs, n_uniform_inputs, n_data, drawn;
    uint32_t reserved1;
    const bsvi_mvn_insn* code;                 /* host: the expression A (the last instruction is its value); MAT: [n_mats][rows][cols] */
    const float* mats;                         /* host [n_mats][rows][cols] */
    const bsvi_uniform_entry* uniform_inputs;  /* host [n_uniform_inputs], parameter-sourced */
    const float* data_mean;                    /* host [n_data][rows * cols] */
    const float* data_scale;                   /* host [n_data][rows * cols] (drawn != 0) */
    float weight;                              /* of the drawn node's log-probability in f */
    uint32_t reserved2;
} bsvi_reduce_desc;
typedef struct bsvi_reduce_args {
    uint32_t struct_size, reserved0;           /* sizeof(bsvi_reduce_args) */
    const float* params_dev;
    const float* samples_dev;                  /* [rows][n_local]: the slot values of the draw (samples_out of a bsvi_elbo_fwd_bwd call) */
    float* rows_out_dev;
    uint32_t n_samples_local, reserved1;
    uint32_t input_rows[8];                    /* row of scalar input k < n_slot_inputs in samples_dev */
    const float* data_dev;                     /* [n_data][rows * cols]: the caller's value of the drawn node, or NULL: drawn from (seed, offset) */
    uint64_t seed, offset;
    void* stream;
} bsvi_reduce_args;
typedef struct bsvi_reduce bsvi_reduce;
int bsvi_reduce_create(const bsvi_reduce_desc* desc, bsvi_reduce** out);
void bsvi_reduce_destroy(bsvi_reduce* r);
uint32_t bsvi_reduce_rows_out(const bsvi_reduce_desc* desc);
int bsvi_reduce_eval(bsvi_reduce* r, const bsvi_reduce_args* args);
/* the generated translation unit (host only, like bsvi_mvn_source): byte count including the terminator, 0 on error */
size_t bsvi_reduce_source(const bsvi_reduce_desc* desc, char* buf, size_t capacity);

/* =========================================================================================
 * The exchange of the multi-GPU path (the reference is single-process and has none).
 *
 * Samples are sharded over the GPUs of a node; between bsvi_elbo_fwd_bwd (this rank's sums in out_dev[0 .. 4 + P)) and
 * bsvi_finalize_step (the replicated optimizer step) every rank needs the TOTAL of the blocks.  The step of a rank is
 * therefore three C calls on one stream: bsvi_elbo_fwd_bwd, one of the two calls below, bsvi_finalize_step.
 *
 * bsvi_allreduce: in-place sum of buf_dev[0 .. n) over the ranks of an RCCL communicator the host owns (`rccl_comm` is an
 *   ncclComm_t; with PyTorch: ProcessGroupNCCL._comm_ptr()).  The library does not link RCCL: ncclAllReduce is resolved at
 *   the first call from the RCCL instance already loaded in the process (BSVI_RCCL_LIB names another).  Stream-ordered,
 *   capturable into a HIP graph like the launches around it.
 *
 * bsvi_exchange_*: a one-shot direct-write all-reduce for messages of at most 16384 floats (this path's are 47 to ~400):
 *   every rank owns a region of device memory; bsvi_exchange_export gives its HIP IPC handle (bsvi_exchange_handle_bytes
 *   bytes) which the host passes to the other ranks by whatever channel it has; bsvi_exchange_connect takes the handles of
 *   all ranks ([world][handle bytes], rank order) and maps the peers' regions (over xGMI between GPUs).
 *   bsvi_exchange_allreduce is ONE one-workgroup kernel: the rank's vector is written into its slot of every region, a
 *   per-call sequence number is published behind a system-scope release, the kernel waits for the peers' numbers and adds
 *   the slots in rank order — the same association on every rank, so all ranks hold bit-identical totals.  The wait is
 *   bounded (BSVI_EXCHANGE_TIMEOUT_MS, default 2000): a rank whose peers never arrive leaves buf_dev unchanged and raises
 *   the region's abort word — bsvi_exchange_status returns the sequence number of the call that gave up (0: none did).
 *   Nothing traps or hangs.  Every rank must make the same sequence of bsvi_exchange_allreduce calls. */
typedef struct bsvi_exchange bsvi_exchange;
int bsvi_allreduce(void* rccl_comm, float* buf_dev, size_t n, void* stream);
size_t bsvi_exchange_handle_bytes(void);
int bsvi_exchange_create(uint32_t rank, uint32_t world, uint32_t capacity_floats, bsvi_exchange** out);
int bsvi_exchange_export(const bsvi_exchange* x, void* handle_out);
int bsvi_exchange_connect(bsvi_exchange* x, const void* handles);
int bsvi_exchange_allreduce(bsvi_exchange* x, float* buf_dev, uint32_t n, void* stream);
int bsvi_exchange_status(const bsvi_exchange* x);
/* (ABI 11) one all-reduce of 1..64 floats through the region's TAGGED-ENTRY area — the 8-byte `(call number << 32) | value` entries the
 * in-kernel training loop exchanges through (bsvi_train_persistent_exchange), with its numbering, scopes, bounded wait and abort word — so
 * that a host can test that area between its GPUs before it lets a training loop rely on it.  An abandoned call leaves NaN in buf_dev. */
int bsvi_exchange_selftest_tagged(bsvi_exchange* x, float* buf_dev, uint32_t n, void* stream);
void bsvi_exchange_destroy(bsvi_exchange* x);

/* (ABI 11) The minibatch data path of the scalar engine (standard_variables.py:71-112, distributions.py:393-473: EmpiricalVariable /
 * RandomIndices / EmpiricalDistribution._get_sample): writes batch_size rows of dataset_dev [dataset_size][row_floats] into dst_dev — a
 * stretch of the observation buffer bsvi_elbo_args::obs_dev that the program reads as ordinary observations — in front of an evaluation.
 * Row i is indices_dev[i] when given, else position i of a keyed bijection of [0, dataset_size) (distinct rows: sampling without
 * replacement, np.random.choice(replace=False) at distributions.py:438) that depends on (seed, offset) only: the dense-link family
 * draws the same rows for the same pair.  indices_out_dev (or NULL) receives the rows used. */
int bsvi_minibatch_gather(const float* dataset_dev, uint32_t dataset_size, uint32_t row_floats, uint32_t batch_size,
                          const int32_t* indices_dev, uint64_t seed, uint64_t offset, float* dst_dev, int32_t* indices_out_dev,
                          void* stream);

const char* bsvi_last_error(void);
int bsvi_abi_version(void);
/* number of gfx950 devices visible to the HIP runtime (0 if none) */
int bsvi_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* BSVI_H */
