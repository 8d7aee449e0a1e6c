// collective.hip — the ONE exchange of the multi-GPU path behind the C ABI (include/bsvi.h).
//
// Monte-Carlo samples are sharded over the GPUs of a node; per step every rank holds [4 + P] sums (loss sum, non-finite
// count, gradient sums) and all ranks need their total.  Two ways to get it, both callable from any host language between
// bsvi_elbo_fwd_bwd and bsvi_finalize_step:
//
//   bsvi_allreduce         RCCL's ncclAllReduce on a communicator the host owns (torch.distributed's, or its own).  The
//                          library does not link RCCL: the entry point is resolved at the first call from the RCCL
//                          instance already loaded in the process, so communicator and code come from the same instance.
//
//   bsvi_exchange_*        a one-shot direct-write all-reduce for the small messages of this path (188 B at BASELINE
//                          config 1, 1.6 KB at config 3; RCCL's ring is latency-bound there): every rank owns a region
//                          of device memory that its peers map through HIP IPC (over xGMI between GPUs); ONE kernel per
//                          call writes the rank's vector into its slot of every peer's region, publishes a per-call
//                          sequence number behind a system-scope release, waits for the peers' numbers (bounded: a rank
//                          that never arrives raises the region's abort word, nothing traps or hangs), and adds the
//                          slots in rank order — the same association on every rank, so the totals are bit-identical
//                          everywhere and the replicated optimizer steps stay in lockstep.
//
// The reference has no collectives (it is a single-process PyTorch-CPU program); the partitioning is SURVEY §8e.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "bsvi.h"
#include "bsvi_internal.h"
#include "spec_args.h"      // the region layout, shared with the generated kernels that exchange inside their training loop

// ---------------------------------------------------------------------------------------------------------------
//  RCCL through the instance the process already has
// ---------------------------------------------------------------------------------------------------------------
namespace {

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*nccl_error_fn)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;        // rccl.h: ncclFloat32 = 7, ncclSum = 0

std::mutex g_rccl_mu;
nccl_allreduce_fn g_allreduce = nullptr;
nccl_error_fn g_error_string = nullptr;
std::string g_rccl_origin;

int resolve_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_allreduce) return BSVI_OK;
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    void* lib = nullptr;
    g_rccl_origin = "global scope";
    if (!sym) {
        // loaded, but not in the global scope (a Python extension's dependency): by its soname, without loading a second copy
        const char* names[] = {getenv("BSVI_RCCL_LIB"), "librccl.so.1", "librccl.so", "libnccl.so.2"};
        for (const char* name : names) {
            if (!name || !name[0]) continue;
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (lib) { g_rccl_origin = name; break; }
        }
        if (!lib && getenv("BSVI_RCCL_LIB")) {             // the host asked for a specific library: load it
            lib = dlopen(getenv("BSVI_RCCL_LIB"), RTLD_NOW);
            if (lib) g_rccl_origin = getenv("BSVI_RCCL_LIB");
        }
        if (lib) sym = dlsym(lib, "ncclAllReduce");
    }
    if (!sym)
        return bsvi_fail(BSVI_ERR_UNSUPPORTED, "bsvi_allreduce: no RCCL instance is loaded in this process (create the communicator "
                                               "first, or name the library in BSVI_RCCL_LIB)");
    g_allreduce = (nccl_allreduce_fn)sym;
    void* es = lib ? dlsym(lib, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString");
    g_error_string = (nccl_error_fn)es;
    return BSVI_OK;
}

}  // namespace

extern "C" int bsvi_allreduce(void* rccl_comm, float* buf_dev, size_t n, void* stream) {
    if (!rccl_comm || !buf_dev) return bsvi_fail(BSVI_ERR_INVALID, "bsvi_allreduce: null communicator or buffer");
    if (n == 0) return BSVI_OK;
    int rc = resolve_rccl();
    if (rc) return rc;
    const int res = g_allreduce(buf_dev, buf_dev, n, kNcclFloat32, kNcclSum, rccl_comm, (hipStream_t)stream);
    if (res != 0)
        return bsvi_fail(BSVI_ERR_HIP, std::string("ncclAllReduce: ") + (g_error_string ? g_error_string(res) : "error ") + " (" + std::to_string(res) + ")");
    return BSVI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
//  the one-shot direct-write exchange
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t kMaxRanks = bsvi::XCHG_MAX_RANKS;           // one node: 8 GPUs, 7 xGMI peers each
constexpr uint32_t kFlagStride = bsvi::XCHG_FLAG_STRIDE;       // a flag per 64-byte line
constexpr uint32_t kHeaderWords = bsvi::XCHG_HEADER_WORDS;     // flags [8 x 16] | abort | timeouts | calls | ... (1 KB)
constexpr uint32_t kCallsWord = bsvi::XCHG_CALLS_WORD;

struct XArgs {
    unsigned char* peer[kMaxRanks];         // every rank's region as mapped here (peer[rank] = this rank's own)
    float* buf;
    uint32_t n, capacity, rank, world;
    unsigned long long timeout_ticks;       // of the 100 MHz wall clock
};

__device__ __forceinline__ float* slot_of(unsigned char* region, uint32_t parity, uint32_t r, uint32_t capacity, uint32_t world) {
    return reinterpret_cast<float*>(region + kHeaderWords * 4) + ((size_t)parity * world + r) * capacity;
}

// ONE workgroup.  Stores to the peers and the loads of what they wrote are system-scope atomics (relaxed): they bypass this
// GPU's caches, so a slot written by a peer two calls ago cannot be served stale from L2.
//
// Failure is sticky and shared: a rank that gives up waiting raises the abort word of EVERY region; a rank that finds its
// abort word raised — at the start of a call, while it waits, or after the wait — abandons the call too.  An abandoned call
// leaves NaN in buf[0] (the loss sum) and buf[1] (the non-finite count): bsvi_finalize_step then sees a non-finite loss and
// skips the optimizer step, on every rank, so the replicated parameters cannot drift apart silently, and the NaN in the loss
// curve plus bsvi_exchange_status tell the host.  (Before: the rank that gave up kept its own partial sums — finite, and
// wrong — while a late peer completed the call normally.)
__global__ __launch_bounds__(256) void exchange_kernel(const XArgs A) {
    const uint32_t tid = threadIdx.x;
    __shared__ uint32_t gave_up, seq_s;
    uint32_t* mine = reinterpret_cast<uint32_t*>(A.peer[A.rank]);
    uint32_t* const abort_word = mine + kMaxRanks * kFlagStride;
    // the call's sequence number lives in the region (word kCallsWord, touched by this rank's kernels only): the launch is the
    // same every time, so a HIP graph that captured it replays correctly
    if (tid == 0) {
        gave_up = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u ? 1u : 0u;
        seq_s = mine[kCallsWord] + 1u;
    }
    __syncthreads();
    const uint32_t seq = seq_s, parity = seq & 1u;
    // 1. this rank's vector into its slot of every region (its own included)
    for (uint32_t i = tid; i < A.n; i += 256) {
        const float v = A.buf[i];
        for (uint32_t p = 0; p < A.world; ++p)
            __hip_atomic_store(slot_of(A.peer[p], parity, A.rank, A.capacity, A.world) + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    // 2. publish: "rank's slot of call seq is complete" in every region
    if (tid < A.world)
        __hip_atomic_store(reinterpret_cast<uint32_t*>(A.peer[tid]) + A.rank * kFlagStride, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid == 0) mine[kCallsWord] = seq;
    // 3. wait for every rank's number in this rank's region — bounded, and ended by anybody's abort
    if (tid < A.world && !gave_up) {
        const unsigned long long t0 = wall_clock64();
        // (sequence numbers only grow; a peer may already be one call ahead)
        while ((int32_t)(__hip_atomic_load(mine + tid * kFlagStride, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { gave_up = 1; break; }
            if (wall_clock64() - t0 > A.timeout_ticks) {
                gave_up = 1;
                for (uint32_t p = 0; p < A.world; ++p)          // the abort word of every region: the call that gave up
                    __hip_atomic_store(reinterpret_cast<uint32_t*>(A.peer[p]) + kMaxRanks * kFlagStride, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                atomicAdd(abort_word + 1, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    if (tid == 0 && !gave_up && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) gave_up = 1;
    __syncthreads();
    if (gave_up) {                           // abandoned: poison the sums (see above); the host reads bsvi_exchange_status
        if (tid < 2 && tid < A.n) A.buf[tid] = __int_as_float(0x7fc00000);
        return;
    }
    // 4. the total, slots added in rank order
    unsigned char* region = A.peer[A.rank];
    for (uint32_t i = tid; i < A.n; i += 256) {
        float s = 0.0f;
        for (uint32_t r = 0; r < A.world; ++r)
            s += __hip_atomic_load(slot_of(region, parity, r, A.capacity, A.world) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        A.buf[i] = s;
    }
}

// The in-loop exchange's entry area (spec_main.h: spec_xput / spec_xget — tagged 8-byte entries `(call number << 32) | value bits`, value
// and "it is there" in one store, no flag) exercised by itself: ONE wave, lane l owns element l of the message, stores its entry into
// its row of every region and polls the entries of every rank in its own region until all carry the call's number; the entries are
// added in rank order.  Same layout, same scopes, same bounded wait and abort word as the generated kernels — what `Exchange.self_test`
// runs before a process lets its training loop exchange inside the kernel (the self-test covered the flagged slots
// of exchange_kernel only).  The in-loop exchanges number their calls in word kCallsWord + 1 of the rank's own region.
__device__ __forceinline__ unsigned long long* entry_of(unsigned char* region, uint32_t parity, uint32_t r, uint32_t capacity, uint32_t world) {
    return reinterpret_cast<unsigned long long*>(region + kHeaderWords * 4 + (size_t)2 * world * capacity * 4) + ((size_t)parity * world + r) * capacity;
}
__global__ __launch_bounds__(64) void tagged_selftest_kernel(const XArgs A) {
    const uint32_t l = threadIdx.x;
    uint32_t* mine = reinterpret_cast<uint32_t*>(A.peer[A.rank]);
    uint32_t* const abort_word = mine + kMaxRanks * kFlagStride;
    const uint32_t seq = __builtin_amdgcn_readfirstlane(mine[kCallsWord + 1]) + 1u, parity = seq & 1u;
    const bool live = l < A.n;
    const unsigned long long tag = (unsigned long long)seq << 32;
    const float v = live ? A.buf[l] : 0.0f;
    if (live)
        for (uint32_t p = 0; p < A.world; ++p)
            __hip_atomic_store(entry_of(A.peer[p], parity, A.rank, A.capacity, A.world) + l, tag | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned char* const region = A.peer[A.rank];
    unsigned long long t0 = 0;
    bool gave_up = false;
    float total = 0.0f;
    for (uint32_t round = 0;; ++round) {
        bool all = true;
        total = 0.0f;
        for (uint32_t r = 0; r < A.world; ++r) {
            const unsigned long long e = live ? __hip_atomic_load(entry_of(region, parity, r, A.capacity, A.world) + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : tag;
            all = all && (uint32_t)(e >> 32) == seq;
            total += __uint_as_float((uint32_t)e);
        }
        if (__all((int)all)) break;
        if (round == 0u) t0 = wall_clock64();
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { gave_up = true; break; }
        if (wall_clock64() - t0 > A.timeout_ticks) {
            gave_up = true;
            if (l < A.world) __hip_atomic_store(reinterpret_cast<uint32_t*>(A.peer[l]) + kMaxRanks * kFlagStride, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (l == 0u) atomicAdd(abort_word + 1, 1u);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    if (l == 0u) mine[kCallsWord + 1] = seq;
    if (gave_up || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) total = __int_as_float(0x7fc00000);
    if (live) A.buf[l] = total;
}

}  // namespace

struct bsvi_exchange {
    uint32_t rank = 0, world = 1, capacity = 0, seq = 0;
    size_t bytes = 0;
    unsigned char* region = nullptr;            // this rank's (device memory)
    unsigned char* peer[kMaxRanks] = {};
    bool opened[kMaxRanks] = {};
    bool connected = false;
    unsigned long long timeout_ticks = 0;
    bsvi::SpecExchange* desc_dev = nullptr;     // the exchange as a kernel that runs it inside its own loop reads it (spec_main.h)
};

// device-resident copy of (peers, rank, world, capacity, timeout): written once, when every peer's region is mapped
static int publish_descriptor(bsvi_exchange* x) {
    bsvi::SpecExchange d;
    memset(&d, 0, sizeof d);
    for (uint32_t r = 0; r < x->world; ++r) d.peer[r] = x->peer[r];
    d.rank = x->rank; d.world = x->world; d.capacity = x->capacity; d.timeout_ticks = x->timeout_ticks;
    if (!x->desc_dev && hipMalloc((void**)&x->desc_dev, sizeof d) != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, "bsvi_exchange: descriptor allocation failed");
    if (hipMemcpy(x->desc_dev, &d, sizeof d, hipMemcpyHostToDevice) != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, "bsvi_exchange: descriptor upload failed");
    return BSVI_OK;
}

// (bsvi_internal.h) the descriptor for a message of n floats, or null with the reason in bsvi_last_error
const void* bsvi_exchange_descriptor(bsvi_exchange* x, uint32_t n) {
    if (!x || !x->connected) { (void)bsvi_fail(BSVI_ERR_INVALID, "the exchange's peers are not connected yet"); return nullptr; }
    if (n > x->capacity) { (void)bsvi_fail(BSVI_ERR_INVALID, "message longer than the exchange's capacity"); return nullptr; }
    if (!x->desc_dev && publish_descriptor(x) != BSVI_OK) return nullptr;
    return x->desc_dev;
}

extern "C" size_t bsvi_exchange_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

extern "C" int bsvi_exchange_create(uint32_t rank, uint32_t world, uint32_t capacity_floats, bsvi_exchange** out) {
    if (!out || world < 1 || world > kMaxRanks || rank >= world || capacity_floats == 0 || capacity_floats > (1u << 14))
        return bsvi_fail(BSVI_ERR_INVALID, "bsvi_exchange_create: 1..8 ranks and 1..16384 floats (the one-shot form is for small messages; "
                                           "larger ones belong to bsvi_allreduce)");
    auto* x = new bsvi_exchange();
    x->rank = rank; x->world = world; x->capacity = (capacity_floats + 63u) / 64u * 64u;
    // header | the flagged slots of exchange_kernel | the tagged 8-byte entries of the in-loop exchange (spec_args.h)
    x->bytes = (size_t)kHeaderWords * 4 + 2 * (size_t)world * x->capacity * sizeof(float) + 2 * (size_t)world * x->capacity * 8;
    // fine-grained memory: uncached across devices — what a region written by peers on OTHER GPUs needs while the spinning
    // kernel is resident.  Coarse-grained device memory gives no such guarantee: it serves one rank, or ranks that share a GPU
    // (the single-GPU tests say so with BSVI_EXCHANGE_SAME_DEVICE=1); otherwise the caller is told to use RCCL.
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, x->bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        const char* same = getenv("BSVI_EXCHANGE_SAME_DEVICE");
        if (world > 1 && !(same && same[0] == '1')) {
            delete x;
            return bsvi_fail(BSVI_ERR_UNSUPPORTED, "bsvi_exchange_create: no fine-grained device memory on this device — the one-shot exchange "
                                                   "cannot guarantee visibility between GPUs; use bsvi_allreduce (RCCL)");
        }
        if (hipMalloc(&p, x->bytes) != hipSuccess) { delete x; return bsvi_fail(BSVI_ERR_HIP, "bsvi_exchange_create: device allocation failed"); }
    }
    x->region = (unsigned char*)p;
    if (hipMemset(x->region, 0, x->bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(x->region);
        delete x;
        return bsvi_fail(BSVI_ERR_HIP, "bsvi_exchange_create: clearing the region failed");
    }
    x->peer[rank] = x->region;
    const char* t = getenv("BSVI_EXCHANGE_TIMEOUT_MS");
    const double ms = t ? atof(t) : 2000.0;
    x->timeout_ticks = (unsigned long long)((ms > 0 ? ms : 2000.0) * 1e5);      // wall_clock64 ticks at 100 MHz
    if (world == 1) x->connected = true;
    *out = x;
    return BSVI_OK;
}

extern "C" int bsvi_exchange_export(const bsvi_exchange* x, void* handle_out) {
    if (!x || !handle_out) return bsvi_fail(BSVI_ERR_INVALID, "null argument");
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, x->region);
    if (e != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e));
    memcpy(handle_out, &h, sizeof h);
    return BSVI_OK;
}

extern "C" int bsvi_exchange_connect(bsvi_exchange* x, const void* handles) {
    if (!x || !handles) return bsvi_fail(BSVI_ERR_INVALID, "null argument");
    if (x->connected) return BSVI_OK;
    for (uint32_t r = 0; r < x->world; ++r) {
        if (r == x->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)r * sizeof h, sizeof h);
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, std::string("hipIpcOpenMemHandle (rank ") + std::to_string(r) + "): " + hipGetErrorString(e));
        x->peer[r] = (unsigned char*)p;
        x->opened[r] = true;
    }
    x->connected = true;
    return publish_descriptor(x);
}

extern "C" int bsvi_exchange_allreduce(bsvi_exchange* x, float* buf_dev, uint32_t n, void* stream) {
    if (!x || !buf_dev) return bsvi_fail(BSVI_ERR_INVALID, "null argument");
    if (!x->connected) return bsvi_fail(BSVI_ERR_INVALID, "bsvi_exchange_allreduce: the peers' regions are not connected yet");
    if (n > x->capacity) return bsvi_fail(BSVI_ERR_INVALID, "bsvi_exchange_allreduce: message longer than the exchange's capacity");
    if (n == 0) return BSVI_OK;
    XArgs A{};
    for (uint32_t r = 0; r < x->world; ++r) A.peer[r] = x->peer[r];
    A.buf = buf_dev; A.n = n; A.capacity = x->capacity; A.rank = x->rank; A.world = x->world;
    ++x->seq;                                   // (host-side count only: the kernel numbers its calls in the region itself)
    A.timeout_ticks = x->timeout_ticks;
    hipLaunchKernelGGL(exchange_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, A);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, std::string("exchange_kernel launch: ") + hipGetErrorString(e));
    return BSVI_OK;
}

// one all-reduce of up to 64 floats through the TAGGED ENTRY area (the in-loop exchange's: tagged_selftest_kernel above); an abandoned
// call leaves NaN in every element and is sticky like any other (bsvi_exchange_status)
extern "C" int bsvi_exchange_selftest_tagged(bsvi_exchange* x, float* buf_dev, uint32_t n, void* stream) {
    if (!x || !buf_dev) return bsvi_fail(BSVI_ERR_INVALID, "null argument");
    if (!x->connected) return bsvi_fail(BSVI_ERR_INVALID, "bsvi_exchange_selftest_tagged: the peers' regions are not connected yet");
    if (n == 0 || n > 64 || n > x->capacity) return bsvi_fail(BSVI_ERR_INVALID, "bsvi_exchange_selftest_tagged: 1..64 floats");
    XArgs A{};
    for (uint32_t r = 0; r < x->world; ++r) A.peer[r] = x->peer[r];
    A.buf = buf_dev; A.n = n; A.capacity = x->capacity; A.rank = x->rank; A.world = x->world;
    A.timeout_ticks = x->timeout_ticks;
    hipLaunchKernelGGL(tagged_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bsvi_fail(BSVI_ERR_HIP, std::string("tagged_selftest_kernel launch: ") + hipGetErrorString(e));
    return BSVI_OK;
}

// 0: every call so far met its peers; otherwise the sequence number of the last call that gave up waiting (synchronises)
extern "C" int bsvi_exchange_status(const bsvi_exchange* x) {
    if (!x) return -1;
    uint32_t words[2] = {0, 0};
    if (hipMemcpy(words, x->region + (size_t)kMaxRanks * kFlagStride * 4, sizeof words, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)words[0];
}

extern "C" void bsvi_exchange_destroy(bsvi_exchange* x) {
    if (!x) return;
    for (uint32_t r = 0; r < x->world; ++r)
        if (x->opened[r] && x->peer[r]) (void)hipIpcCloseMemHandle(x->peer[r]);
    if (x->region) (void)hipFree(x->region);
    if (x->desc_dev) (void)hipFree(x->desc_dev);
    delete x;
}
