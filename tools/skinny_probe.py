"""Timing of single launches of the narrow-layer kernels (bsvi_debug_gemm with a side <= 8) at the cfg 5 shapes.
usage (GPU box): python3 tools/skinny_probe.py"""
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
cases = [("heads fwd   x W^T  (NT)", 0, (R, 4, 512)), ("dec l1 fwd  z W^T  (NT)", 0, (R, 512, 2)),
         ("dec l1 bwd  dh W   (NN)", 1, (R, 2, 512)), ("heads bwd   dy W   (NN)", 1, (R, 512, 4)),
         ("dec l1 wgt  dh^T z (TN)", 2, (512, 2, R)), ("heads wgt   dy^T h (TN)", 2, (4, 512, R))]
for name, mode, (M, N, K) in cases:
    pad = lambda n: (n + 3) // 4 * 4
    if mode == 0:
        A, B, Cm = torch.randn(M, pad(K), device=dev), torch.randn(N, K, device=dev), torch.zeros(M, pad(N), device=dev)
        bias = torch.randn(N, device=dev)
        fn = lambda: lib.bsvi_debug_gemm(0, ptr(A), ptr(B), ptr(Cm), None, M, N, K, pad(K), K, pad(N), ptr(bias), 0, 1, 0.0, 0, None)
        moved = 4 * (M * K + M * N)
    elif mode == 1:
        A, B, Cm = torch.randn(M, pad(K), device=dev), torch.randn(K, N, device=dev), torch.zeros(M, pad(N), device=dev)
        Y = torch.randn(M, pad(N), device=dev)
        fn = lambda: lib.bsvi_debug_gemm(1, ptr(A), ptr(B), ptr(Cm), None, M, N, K, pad(K), N, pad(N), ptr(Y), pad(N), 1, 0.0, 0, None)
        moved = 4 * (M * K + 2 * M * N)
    else:
        A, B, Cm = torch.randn(K, pad(M), device=dev), torch.randn(K, pad(N), device=dev), torch.zeros(M, N, device=dev)
        fn = lambda: lib.bsvi_debug_gemm(2, ptr(A), ptr(B), ptr(Cm), None, M, N, K, pad(M), pad(N), N, None, 0, 0, 0.0, 0, None)
        moved = 4 * (K * M + K * N)
    us = timed(fn)
    print("%-26s M=%-6d N=%-4d K=%-6d %7.1f us  %5.2f TB/s" % (name, M, N, K, us, moved / us / 1e6))
