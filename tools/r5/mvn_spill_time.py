"""round 5: the batched multivariate-normal kernel at sizes on both sides of the LDS limit (192): us per launch at 300 samples.
Up to 192 a sample's matrix is in LDS; beyond it in a block of device memory (MVN_SPILL, csrc/mvn_kernel.h).
usage: python tools/r5/mvn_spill_time.py [N]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from test_gpu_mvn import gp_node
from brancher_amd import native

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lib = native.load()
dev = torch.device("cuda:0")
for D in (64, 128, 160, 192, 193, 200, 256, 384, 512, 768, 1024):
    node, sq, eye = gp_node(D, False, weight=1.0, seed=D, jitter=5e-2)
    d, keep = native.mvn_desc(node)
    h = C.c_void_p()
    native.check(lib.bsvi_mvn_create(C.byref(d), C.byref(h)))
    n_out = int(lib.bsvi_mvn_rows_out(C.byref(d)))
    samples = torch.zeros(8, N)
    samples[3] = torch.exp(-0.4 + 0.25 * torch.randn(N))
    samples_d, params_d = samples.to(dev), torch.tensor([0.0, 0.0, 0.9, 0.0], device=dev)
    out = torch.zeros(n_out, N, device=dev)
    args = native.MvnArgs(params_dev=params_d.data_ptr(), samples_dev=samples_d.data_ptr(), rows_out_dev=out.data_ptr(),
                          n_samples_local=N, value_row0=5, stream=None)
    args.input_rows[0] = 3
    t0 = time.time()
    native.check(lib.bsvi_mvn_eval(h, C.byref(args)))
    torch.cuda.synchronize()
    first = time.time() - t0
    reps = 20 if D <= 256 else 5
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        native.check(lib.bsvi_mvn_eval(h, C.byref(args)))
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    flops = N * (D ** 3 / 3 + D ** 3 / 3 + D ** 3 / 3)          # factorisation, inverse, X^T X
    print("D %4d  %s  %9.1f us per launch at %d samples  (%.2f TFLOP/s f32; first call incl. compile %.1f s)  finite %s"
          % (D, "LDS   " if D <= 192 else "memory", us, N, flops / us / 1e6, first, bool(torch.isfinite(out).all())), flush=True)
    lib.bsvi_mvn_destroy(h)
