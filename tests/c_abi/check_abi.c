/* The C ABI driven by something that is not brancher_amd/native.py: a C99 program compiled against include/bsvi.h (and the HIP
 * runtime API for device memory) that loads tests/c_abi/readme_ar_T20_N300.blob — the lowered README autoregressive model
 * (BASELINE config 1), the noise of the reference fixture and the REFERENCE's loss and gradients for it — creates the
 * program, runs bsvi_elbo_fwd_bwd + bsvi_finalize on the supplied noise and compares (1e-5 of the loss, 1e-5 of the largest
 * gradient: BASELINE.json north_star).  Then: the Philox path gives a finite loss, and a struct of another size is refused.
 * The seam it stands in for: brancher/inference.py:114-126 (InferenceMethod.compute_loss), brancher/variables.py:843-870.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tests/c_abi/check_abi.c -o check_abi \
 *       -Lbrancher_amd -lbsvi -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/brancher_amd -Wl,-rpath,/opt/rocm/lib
 *   ./check_abi tests/c_abi/readme_ar_T20_N300.blob            (built and run by tests/test_gpu_c_abi.py) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "bsvi.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_BSVI(x) do { int rc_ = (x); if (rc_ != BSVI_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, bsvi_last_error()); return 3; } } while (0)

typedef struct blob_array { uint64_t bytes; void* data; } blob_array;

static int read_array(FILE* f, blob_array* a) {
    if (fread(&a->bytes, 8, 1, f) != 1) return -1;
    const size_t padded = (size_t)((a->bytes + 7) / 8 * 8);
    a->data = malloc(padded ? padded : 8);
    if (padded && fread(a->data, 1, padded, f) != padded) return -1;
    return 0;
}

static void* to_device(const void* host, size_t bytes) {
    void* dev = NULL;
    if (hipMalloc(&dev, bytes ? bytes : 4) != hipSuccess) return NULL;
    if (bytes && hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return dev;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <blob>\n", argv[0]); return 1; }
    /* the binding's view of every struct against the library's */
    const size_t mine[] = {sizeof(bsvi_uniform_entry), sizeof(bsvi_record), sizeof(bsvi_program_desc),
        sizeof(bsvi_elbo_args), sizeof(bsvi_opt_cfg), sizeof(bsvi_dense_desc), sizeof(bsvi_dense_args), sizeof(bsvi_mlp_layer),
        sizeof(bsvi_amort_desc), sizeof(bsvi_amort_args), sizeof(bsvi_mvn_insn), sizeof(bsvi_mvn_desc), sizeof(bsvi_mvn_args),
        sizeof(bsvi_bnn_layer), sizeof(bsvi_bnn_desc), sizeof(bsvi_bnn_args), sizeof(bsvi_reduce_desc), sizeof(bsvi_reduce_args)};
    _Static_assert(sizeof mine / sizeof mine[0] == BSVI_SK_COUNT, "a struct kind of bsvi.h is missing from this list");
    for (int k = 0; k < BSVI_SK_COUNT; ++k)
        if (bsvi_sizeof(k) != mine[k]) { fprintf(stderr, "struct kind %d: header %zu bytes, library %zu\n", k, mine[k], bsvi_sizeof(k)); return 4; }
    if (bsvi_abi_version() != BSVI_ABI_VERSION) { fprintf(stderr, "ABI %d != %d\n", bsvi_abi_version(), BSVI_ABI_VERSION); return 4; }

    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    char magic[8];
    uint32_t head[2], sc[12];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "BSVIBLOB", 8) || fread(head, 4, 2, f) != 2 || fread(sc, 4, 12, f) != 12) {
        fprintf(stderr, "not a blob\n"); return 1;
    }
    if (head[1] != BSVI_ABI_VERSION) { fprintf(stderr, "blob written for ABI %u: rerun tests/c_abi/make_blob.py\n", head[1]); return 1; }
    blob_array a[10];
    for (int i = 0; i < 10; ++i) if (read_array(f, &a[i])) { fprintf(stderr, "truncated blob\n"); return 1; }
    fclose(f);
    const uint32_t n_params = sc[0], n_noise = sc[4], n_samples = sc[10];

    bsvi_program_desc d;
    memset(&d, 0, sizeof d);
    d.struct_size = sizeof d;
    d.abi_version = BSVI_ABI_VERSION;
    d.n_params = sc[0]; d.n_consts = sc[1]; d.n_obs = sc[2]; d.n_slots = sc[3]; d.n_noise = sc[4]; d.n_uniform = sc[5];
    d.n_uniform_grad = sc[6]; d.n_records = sc[7]; d.n_code = sc[8]; d.estimator = sc[9];
    d.uniform = (const bsvi_uniform_entry*)a[0].data; d.records = (const bsvi_record*)a[1].data; d.code = (const uint32_t*)a[2].data;
    d.consts = (const float*)a[3].data; d.param_uniform_ptr = (const uint32_t*)a[4].data; d.param_uniform_idx = (const uint32_t*)a[5].data;
    if (a[0].bytes != (uint64_t)d.n_uniform * sizeof(bsvi_uniform_entry) || a[1].bytes != (uint64_t)d.n_records * sizeof(bsvi_record) ||
        a[2].bytes != (uint64_t)d.n_code * 32 || a[6].bytes != (uint64_t)n_params * 4 || a[8].bytes != (uint64_t)n_noise * n_samples * 4 ||
        a[9].bytes != (uint64_t)(1 + n_params) * 4) { fprintf(stderr, "blob tables do not match their counts\n"); return 1; }

    if (bsvi_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 5; }
    bsvi_program* prog = NULL;
    {   /* a descriptor from "another revision of the header" */
        bsvi_program_desc stale = d;
        stale.struct_size -= 8;
        if (bsvi_program_create(&stale, &prog) != BSVI_ERR_INVALID || prog) { fprintf(stderr, "a short bsvi_program_desc was accepted\n"); return 6; }
    }
    CHECK_BSVI(bsvi_program_create(&d, &prog));
    const size_t ws_bytes = bsvi_workspace_bytes(prog, n_samples);
    void *params = to_device(a[6].data, a[6].bytes), *obs = to_device(a[7].data, a[7].bytes), *noise = to_device(a[8].data, a[8].bytes);
    void *out = to_device(NULL, 0), *ws = NULL;
    if (out) (void)hipFree(out);
    CHECK_HIP(hipMalloc(&out, (BSVI_OUT_HEADER + n_params) * 4));
    CHECK_HIP(hipMalloc(&ws, ws_bytes ? ws_bytes : 4));
    if (!params || !obs || !noise) { fprintf(stderr, "device allocation failed\n"); return 2; }

    bsvi_elbo_args args;
    memset(&args, 0, sizeof args);
    args.struct_size = sizeof args;
    args.params_dev = (const float*)params; args.obs_dev = (const float*)obs; args.noise_dev = (const float*)noise;
    args.n_samples_local = n_samples; args.n_samples_global = n_samples; args.out_dev = (float*)out; args.workspace_dev = ws;
    CHECK_BSVI(bsvi_elbo_fwd_bwd(prog, &args));
    CHECK_BSVI(bsvi_finalize(prog, (float*)out, n_samples, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float* host = (float*)malloc((BSVI_OUT_HEADER + n_params) * 4);
    CHECK_HIP(hipMemcpy(host, out, (BSVI_OUT_HEADER + n_params) * 4, hipMemcpyDeviceToHost));
    const float* ref = (const float*)a[9].data;
    double gmax = 0.0, gerr = 0.0;
    for (uint32_t i = 0; i < n_params; ++i) {
        const double r = ref[1 + i], e = fabs((double)host[BSVI_OUT_HEADER + i] - r);
        if (fabs(r) > gmax) gmax = fabs(r);
        if (e > gerr) gerr = e;
    }
    const double lerr = fabs((double)host[2] - ref[0]);
    printf("loss %.6f (reference %.6f, |diff| %.2e), finite flag %g, largest gradient error %.2e of scale %.3e\n", host[2], ref[0], lerr, host[3], gerr, gmax);
    if (host[3] != 1.0f || !(lerr <= 1e-5 * fabs(ref[0])) || !(gerr <= 1e-5 * gmax)) { fprintf(stderr, "MISMATCH against the reference\n"); return 7; }

    /* the in-kernel Philox draw instead of supplied noise */
    args.noise_dev = NULL; args.seed = 7; args.offset = 3;
    CHECK_BSVI(bsvi_elbo_fwd_bwd(prog, &args));
    CHECK_BSVI(bsvi_finalize(prog, (float*)out, n_samples, NULL));
    CHECK_HIP(hipMemcpy(host, out, 16, hipMemcpyDeviceToHost));
    printf("Philox draw: loss %.6f, finite flag %g\n", host[2], host[3]);
    if (host[3] != 1.0f || !(host[2] == host[2])) return 8;

    /* an argument struct of another size */
    args.struct_size += 8;
    if (bsvi_elbo_fwd_bwd(prog, &args) != BSVI_ERR_INVALID) { fprintf(stderr, "a long bsvi_elbo_args was accepted\n"); return 6; }
    printf("refused: %s\n", bsvi_last_error());
    bsvi_program_destroy(prog);
    (void)hipFree(params); (void)hipFree(obs); (void)hipFree(noise); (void)hipFree(out); (void)hipFree(ws);
    printf("C ABI ok\n");
    return 0;
}
