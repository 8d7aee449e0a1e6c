"""Timing of single launches of the amortised path's GEMM (bsvi_debug_gemm) at the cfg 5 layer shapes.
usage (GPU box): python3 tools/gemm_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brancher_amd import native

lib = native.load()
dev = torch.device("cuda:0")
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


R = 25600
for name, mode, (M, N, K) in [("forward  x W^T", 0, (R, 256, 784)), ("forward  x W^T", 0, (R, 512, 256)), ("forward  x W^T", 0, (R, 784, 256)),
                              ("bwd-data dY W", 1, (R, 256, 784)), ("bwd-data dY W", 1, (R, 512, 256)),
                              ("bwd-wgt  dY^T x", 2, (784, 256, R)), ("bwd-wgt  dY^T x", 2, (256, 512, R)), ("bwd-wgt  dY^T x", 2, (256, 784, R))]:
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
    print("%-16s M=%-6d N=%-4d K=%-6d %7.1f us  %5.1f TFLOP/s" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6))
