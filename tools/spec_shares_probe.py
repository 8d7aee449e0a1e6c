"""Probe: one bsvi_elbo_fwd_bwd launch of the program-specialised kernel for cfg 3 (README AR, T = 200, 1024 samples) —
the whole program against ONE of its V shares (every share samples the posterior, takes 1/V of the model's log-prob
records).  Says what a step built from V concurrent share launches could cost."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from brancher_amd import engine, native, workloads as W  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = 1024
model = W.build_readme_ar(W.native_api(), T=T)
c = engine.compile_model(model, None, "pathwise")
c = getattr(c, "__wrapped__", c)
lib = c.lib
dev = c.device


def timed(handle, out, ws, reps=100):
    args = native.ElboArgs.from_buffer_copy(c._elbo_args(N, N, 0, None, 1, 0))
    args.out_dev = out.data_ptr()
    args.workspace_dev = ws.data_ptr()
    args.stream = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(5):
        native.check(lib.bsvi_elbo_fwd_bwd(handle, C.byref(args)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        args.offset = i
        native.check(lib.bsvi_elbo_fwd_bwd(handle, C.byref(args)))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


ws = c.workspace(N)
print("engine", c.native.engine(N, 0))
t0 = time.time()
print("whole program: %.1f us per launch" % timed(c.native.handle, c.out, ws))
for V in (2, 4, 8):
    parts = c.program.shares.get(V)
    if not parts:
        continue
    t0 = time.time()
    sp = native.NativeProgram(c.native._share_program(*parts[0]))
    sl = native.NativeProgram(c.native._share_program(*parts[-1]))
    out = torch.zeros_like(c.out)
    a, b = timed(sp.handle, out, torch.empty_like(ws)), timed(sl.handle, out, torch.empty_like(ws))
    print("V=%d: share 0 %.1f us, last share %.1f us per launch  (engine %s, first use %.1f s)" % (V, a, b, sp.engine(N, 0)["engine"], time.time() - t0))
