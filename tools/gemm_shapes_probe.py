"""Where the f32 MFMA GEMM of the amortised path loses time: the cfg 5 layer shapes (R = 25 600 rows), the same layers at
R = 32 768 (no tile quantisation on 256 CUs) and 4096^3 (the guide's reference shape), per operand layout.
usage (GPU box): python3 tools/gemm_shapes_probe.py [tag]     env: BSVI_GEMM_HALF_BELOW, BSVI_GEMM_PF"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brancher_amd import native

lib = native.load()
dev = torch.device("cuda:0")
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def run(mode, M, N, K):
    if mode == 0:
        A, B, Cm = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.zeros(M, N, device=dev)
        fn = lambda: lib.bsvi_debug_gemm(0, ptr(A), ptr(B), ptr(Cm), None, M, N, K, K, K, N, None, 0, 1, 0.0, 0, None)
    elif mode == 1:
        A, B, Cm = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.zeros(M, N, device=dev)
        Y = torch.randn(M, N, device=dev)
        fn = lambda: lib.bsvi_debug_gemm(1, ptr(A), ptr(B), ptr(Cm), None, M, N, K, K, N, N, ptr(Y), N, 1, 0.0, 0, None)
    else:
        A, B, Cm = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), torch.zeros(M, N, device=dev)
        fn = lambda: lib.bsvi_debug_gemm(2, ptr(A), ptr(B), ptr(Cm), None, M, N, K, M, N, N, None, 0, 0, 0.0, 0, None)
    us = timed(fn)
    return us, 2.0 * M * N * K / us / 1e6


tag = sys.argv[1] if len(sys.argv) > 1 else ""
names = {0: "fwd  x W^T ", 1: "dX   dY W  ", 2: "dW   dY^T x"}
for R in (25600, 32768):
    shapes = [(0, (R, 256, 784)), (0, (R, 512, 256)), (0, (R, 256, 512)), (0, (R, 784, 256)),
              (1, (R, 256, 512)), (1, (R, 512, 256)), (1, (R, 256, 784)),
              (2, (784, 256, R)), (2, (256, 512, R)), (2, (512, 256, R)), (2, (256, 784, R))]
    tot_us = tot_fl = 0.0
    for mode, (M, N, K) in shapes:
        us, tf = run(mode, M, N, K)
        tot_us += us
        tot_fl += 2.0 * M * N * K
        print("%s %s M=%-6d N=%-4d K=%-6d %7.1f us  %6.1f TFLOP/s" % (tag, names[mode], M, N, K, us, tf))
    print("%s R=%d: eleven launches %.1f us, %.1f TFLOP/s" % (tag, R, tot_us, tot_fl / tot_us / 1e6))
for mode in (0, 1, 2):
    us, tf = run(mode, 4096, 4096, 4096)
    print("%s %s 4096^3 %7.1f us  %6.1f TFLOP/s" % (tag, names[mode], us, tf))
