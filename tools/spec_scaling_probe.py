"""Probe: one step launch of the program-specialised kernel for the README AR model at several chain lengths T and sample
counts — where does a T = 200 launch (cfg 3) spend its time: per-node work or the frame (tables, epilogue, row sums)?"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from brancher_amd import engine, native, workloads as W  # noqa: E402


def timed(c, N, reps=100):
    lib, dev = c.lib, c.device
    args = native.ElboArgs.from_buffer_copy(c._elbo_args(N, N, 0, None, 1, 0))
    args.stream = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(5):
        native.check(lib.bsvi_elbo_fwd_bwd(c.native.handle, C.byref(args)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        args.offset = i
        native.check(lib.bsvi_elbo_fwd_bwd(c.native.handle, C.byref(args)))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


for T in (25, 50, 100, 200):
    c = engine.compile_model(W.build_readme_ar(W.native_api(), T=T), None, "pathwise")
    c = getattr(c, "__wrapped__", c)
    row = []
    for N in (64, 256, 1024):
        row.append("N=%d: %.1f us (%s)" % (N, timed(c, N), c.native.engine(N, 0).get("n_blocks")))
    print("T=%d  " % T + "   ".join(row))
