"""Round 6: times the wide products of BASELINE config 5 (25 600 rows) as six products of exact bf16 pieces (bsvi_debug_gemm modes
5 / 6; the split of the weights included) with the kernel the environment selects (BSVI_X6_V=5: round 5's register-staged kernel;
unset: round 6's LDS-DMA kernel).  python3 tools/r6/x6_probe.py"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from brancher_amd import native

lib = native.load()
dev = torch.device("cuda:0")
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
torch.manual_seed(0)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


for (M, N, K) in [(25600, 512, 256), (25600, 256, 512), (25600, 784, 256), (25600, 256, 784), (25600, 512, 784)]:
    A, Bnt, Bnn, bias = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(K, N, device=dev), torch.randn(N, device=dev)
    Y, Cm = torch.randn(M, N, device=dev), torch.zeros(M, N, device=dev)
    fwd = timed(lambda: native.check(lib.bsvi_debug_gemm(5, ptr(A), ptr(Bnt), ptr(Cm), None, M, N, K, K, K, N, ptr(bias), 0, 1, 0.0, 0, None)))
    bwd = timed(lambda: native.check(lib.bsvi_debug_gemm(6, ptr(A), ptr(Bnn), ptr(Cm), None, M, N, K, K, N, N, ptr(Y), N, 1, 0.0, 0, None)))
    fl = 2.0 * M * N * K
    print("M %5d N %4d K %4d   forward %6.1f us (%5.1f TF f32-equivalent)   input gradient %6.1f us (%5.1f TF)" % (M, N, K, fwd, fl / fwd / 1e6, bwd, fl / bwd / 1e6))
