"""round 5: what the event bracket of bench.py's timed region costs a 20-iteration call of cfg 1.  The same call timed on the host
(synchronize, clock, train(20), [record], synchronize, clock) with: no events; torch.cuda.Event created in the region (bench.py until
now); torch events recorded once beforehand; HIP events created with hipEventDisableSystemFence through libamdhip64 on the launch stream.
usage: python tools/r5/event_cost_probe.py"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from brancher_amd import engine, workloads as W  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
run = lambda k: c.train(k, 300, "SGD", seed=0, lr=1e-3)
run(5)
torch.cuda.synchronize()
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    run(200)
    torch.cuda.synchronize()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def hip_event(flags):
    e = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0
    return e


def measure(kind, reps=300):
    wall, dev = [], []
    for _ in range(reps):
        if kind == "torch":
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        elif kind == "torch_warm":
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
        elif kind in ("hip_default", "hip_nofence"):
            a, b = (hip_event(0x0 if kind == "hip_default" else 0x20000000) for _ in range(2))
        torch.cuda.synchronize()
        if kind.startswith("torch"):
            a.record()
        elif kind.startswith("hip"):
            hip.hipEventRecord(a, stream)
        t0 = time.perf_counter()
        run(20)
        if kind.startswith("torch"):
            b.record()
        elif kind.startswith("hip"):
            hip.hipEventRecord(b, stream)
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
        if kind.startswith("torch"):
            dev.append(a.elapsed_time(b))
        elif kind.startswith("hip"):
            ms = ctypes.c_float()
            hip.hipEventElapsedTime(ctypes.byref(ms), a, b)
            dev.append(ms.value)
            hip.hipEventDestroy(a); hip.hipEventDestroy(b)
    wall.sort(); dev.sort()
    med = lambda v: v[len(v) // 2] if v else float("nan")
    print("%-12s wall median %.1f us (best %.1f) = %.0f it/s;  events: median %.1f us" %
          (kind, med(wall) * 1e6, wall[0] * 1e6, 20 / med(wall), med(dev) * 1e3 if dev else float("nan")), flush=True)


for kind in ("none", "torch", "torch_warm", "hip_default", "hip_nofence", "none"):
    measure(kind)
