// bsvi_internal.h — shared between the translation units of libbsvi.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "bsvi.h"

// records the thread-local message returned by bsvi_last_error() and returns `code`
int bsvi_fail(int code, const std::string& msg);
// every descriptor / argument struct starts with struct_size = the CALLER's sizeof (include/bsvi.h): a binding written
// against another header revision is refused instead of being read past its end
#define BSVI_CHECK_STRUCT(ptr, type)                                                                              \
    do {                                                                                                          \
        if ((ptr)->struct_size != sizeof(type))                                                                   \
            return bsvi_fail(BSVI_ERR_INVALID, std::string(#type "::struct_size is ") + std::to_string((ptr)->struct_size) + \
                             ", this library's sizeof is " + std::to_string(sizeof(type)) +                       \
                             ": the binding was written against another revision of include/bsvi.h");             \
    } while (0)
// amort_kernel.hip: C[M][N] = X[rows[m]] W^T on the bf16 matrix cores — X exactly bf16 [..][Kp], Wp the three bf16 pieces
// [3][N][Kp] (hi, mid, lo) of an f32 operand, Kp a multiple of 32
int bsvi_xgemm_nt(const uint16_t* X, const int32_t* rows, const uint16_t* Wp, long plane_stride, float* C, int ldc,
                  int M, int N, int Kp, void* stream);

// amort_kernel.hip: C = A B^T (mode 0: B is [N][K]) or C = A B (mode 1: B is [K][N]) on the f32-input MFMA kernels, any shape
int bsvi_xgemm_nt_t(const uint16_t* X, const int32_t* rows, const uint16_t* Wp, long plane_stride, float* Ct, int ldct,
                    int M, int N, int Kp, void* stream);
int bsvi_gemm_f32(int mode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, void* stream);

namespace bsvi_spec { struct Spec; }

struct bsvi_program {
    bsvi_program_desc d;
    void* dev_blob = nullptr;       // one allocation holding every table
    const uint4* code = nullptr;
    const uint4* aux = nullptr;
    const bsvi_record* records = nullptr;
    const bsvi_uniform_entry* uniform = nullptr;
    const float* consts = nullptr;
    const uint32_t* pu_ptr = nullptr;
    const uint32_t* pu_idx = nullptr;
    int max_lds = 0;
    bool generic = false;   // contains instructions other than NAFF
    const bsvi_program* shares[8] = {};   // bsvi_program_set_shares
    uint32_t n_shares = 0;
    bsvi_spec::Spec* spec = nullptr;      // the program-specialised kernels (specialize.cpp), or null: interpreter only
};

// ---- program specialisation (specialize.cpp): straight-line HIP generated from the instruction stream, compiled
//      with hiprtc on first launch.  The interpreter kernels of elbo_kernel.hip remain the engine for programs the
//      generator declines (very long unrolled streams) and when BSVI_JIT=0.
namespace bsvi_spec {

enum Mode { MODE_SUMS = 0, MODE_STEP = 1, MODE_LOOP = 2 };

// Generate the sources (no compilation; host only).  `desc` must still carry its host tables.  Returns null and
// sets `why` when the program is not specialised.
Spec* create(const bsvi_program_desc& desc, std::string& why);
void destroy(Spec* s);
// device-side tables of the spec (needs a device; called from bsvi_program_create)
int upload(Spec* s);
// the generated translation unit of a variant (0 lean, 1 diagnostic) — tests and bsvi_program_source
const std::string& source(const Spec* s, int variant);
// compile a generated translation unit for gfx950 with hiprtc; `code` receives the code object
int compile(const std::string& src, std::vector<char>& code, std::string& log);
// the same through the caches (this process's, then the one on disk; a miss compiles and fills both).
// origin: 1 hiprtc, 2 process cache, 3 disk cache
int obtain(const std::string& src, std::vector<char>& code, std::string& log, int* origin);
int last_origin();
std::string cache_directory();
std::string compiler_identity();
// bytes of workspace a launch over n_local samples needs behind the interpreter's region
size_t workspace_bytes(const Spec* s, uint32_t n_local);
// true when a launch in `mode` over n_local samples is served by the specialised kernel
bool applies(const Spec* s, uint32_t n_local, int mode);
struct Launch {
    const bsvi_elbo_args* a = nullptr;
    int mode = MODE_SUMS;
    const bsvi_opt_cfg* cfg = nullptr;
    float* params = nullptr;
    float* state = nullptr;
    const uint8_t* mask = nullptr;
    const uint8_t* mask_first = nullptr;
    uint32_t pretraining_iterations = 0, n_iterations = 1;
    float* loss_slot = nullptr;
    float* finite_slot = nullptr;
    void* workspace = nullptr;      // the spec's region of the caller's workspace
    const void* xchg = nullptr;     // loop mode on several ranks: the device-resident descriptor of the exchange (SpecExchange), or null
};
int launch(Spec* s, const bsvi_program* p, const Launch& L);
// geometry of that launch (tests / bench)
void geometry(const Spec* s, uint32_t n_local, int mode, uint32_t* n_blocks, uint32_t* n_threads, uint32_t* lds_bytes);

}  // namespace bsvi_spec

// collective.hip: the device-resident descriptor of a connected exchange for a message of n floats (what a generated kernel
// that exchanges inside its training loop reads: SpecExchange, spec_args.h), or null with the reason in bsvi_last_error
const void* bsvi_exchange_descriptor(bsvi_exchange* x, uint32_t n);
