// spec_args.h — kernel argument block of the program-specialised kernels: ONE definition for the generated device code
// (spec_prelude.h, through hiprtc) and the host launcher (specialize.cpp).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#endif
#include "bsvi.h"

namespace bsvi {

enum { SPEC_MODE_SUMS = 0, SPEC_MODE_STEP = 1, SPEC_MODE_LOOP = 2 };

// The one-shot exchange (collective.hip) as the in-kernel training loop sees it: a rank's region is
//   [header: XCHG_HEADER_WORDS words][slots: 2 parities x world ranks x capacity floats]
//   [entries: 2 parities x world ranks x capacity 8-byte words, (call number << 32) | value bits — the in-loop exchange's]
// header words: flag of rank r at r * XCHG_FLAG_STRIDE (the sequence number of r's last complete call) | abort word at
// XCHG_MAX_RANKS * XCHG_FLAG_STRIDE (+1: how often this rank gave up) | XCHG_CALLS_WORD: this rank's own call count
// (+1: the count of its in-loop exchanges, which number their calls separately).
// ONE definition for the exchange kernel of the library and the generated kernels, which take part in the same sequence.
enum { XCHG_MAX_RANKS = 8, XCHG_FLAG_STRIDE = 16, XCHG_HEADER_WORDS = 256, XCHG_CALLS_WORD = XCHG_MAX_RANKS * XCHG_FLAG_STRIDE + 16 };
struct SpecExchange {                    // device-resident, written once when the peers are connected
    unsigned char* peer[XCHG_MAX_RANKS]; // every rank's region as mapped into this process (peer[rank]: its own)
    uint32_t rank, world, capacity, reserved;
    unsigned long long timeout_ticks;    // of the 100 MHz wall clock
};

// kernel argument block (mirrored by the host in specialize.cpp; plain data, 8-byte aligned pointers first)
struct SpecArgs {
    const bsvi_uniform_entry* uniform;   // [SPEC_N_UNIFORM]
    const float* consts;
    float* params;                       // read by the prologue; written by the fused optimizer step
    const float* obs;
    const float* noise;                  // [n_noise][n_local] or null -> Philox (diagnostic variant only)
    float* samples_out;                  // diagnostic variant only
    float* noise_out;
    float* fvalue_out;
    float* out;                          // [BSVI_OUT_HEADER + n_params]
    float* partials;                     // [grid][2 + SPEC_N_UGRAD]   (grid > 1)
    unsigned int* ticket;                // arrival counter, zero between launches (grid > 1)
    const uint32_t* pu_ptr;              // CSR theta -> uniform entries, entries given as POSITIONS in the order the
    const uint32_t* pu_pos;              //   generated body completes them (spec_du)
    const uint32_t* pu_idx;              //   ... and as uniform-table indices (for the transform)
    float* state;                        // optimizer state [4][n_params]
    const uint8_t* mask;                 // active parameters
    const uint8_t* mask_first;           // ... while iteration <= pretraining_iterations (loop mode)
    float* loss_slot;                    // step: one slot (or null); loop: the loss curve
    float* finite_slot;
    const unsigned long long* offset_dev;   // added to the Philox offset when non-null (bsvi_elbo_args::offset_dev)
    const float* f_weight;               // diagnostic variant: caller weights of grad f_n / grad log q_n (bsvi_elbo_args::f_weight_dev,
    const float* q_weight;               //   q_weight_dev — the second pass of a user-defined gradient estimator), or null
    const SpecExchange* xchg;            // loop mode on several ranks: the sums of every iteration are exchanged INSIDE the loop, or null
    uint32_t n_local, n_global, sample_base, mode;
    uint32_t seed_lo, seed_hi, offset_lo, offset_hi;
    uint32_t n_iterations, pretraining_iterations, n_params, reserved;
    bsvi_opt_cfg cfg;
};

}  // namespace bsvi
