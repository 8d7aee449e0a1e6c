"""round 5: what the host's wait policy costs a 20-iteration call of cfg 1 — hipDeviceScheduleAuto (the default: the waiting thread yields /
blocks) against hipDeviceScheduleSpin, set before the device is initialised.  usage: python tools/r5/sync_policy_probe.py [spin]"""
import ctypes
import os
import sys
import time

import torch

if len(sys.argv) > 1 and sys.argv[1] == "spin":
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", hip.hipSetDeviceFlags(1))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from brancher_amd import engine, workloads as W  # noqa: E402

c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
c.train(5, 300, "SGD", seed=0, lr=1e-3)
torch.cuda.synchronize()
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    c.train(200, 300, "SGD", seed=0, lr=1e-3)
    torch.cuda.synchronize()
best, total = 1e9, 0.0
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c.train(20, 300, "SGD", seed=0, lr=1e-3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    best = min(best, dt)
    total += dt
print("20-iteration call: mean %.1f us, best %.1f us  (%.0f / %.0f it/s)" % (total / 200 * 1e6, best * 1e6, 20 / (total / 200), 20 / best))
