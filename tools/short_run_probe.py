"""Where the time of a SHORT timed region goes (the driver runs bench.py --steps 20 --warmup 5): host time of the train()
call, time until the device is idle again, device time between HIP events.  usage (GPU box): python3 tools/short_run_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from brancher_amd import engine, workloads as W

c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
c.train(200, 300, "SGD", lr=1e-3, seed=0)
torch.cuda.synchronize()
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    c.train(200, 300, "SGD", lr=1e-3, seed=0)
    torch.cuda.synchronize()
for K in (1, 2, 5, 20, 200, 2000):
    rows = []
    for _ in range(30):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        c.train(K, 300, "SGD", lr=1e-3, seed=0)
        t1 = time.perf_counter()
        ev1.record()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6, ev0.elapsed_time(ev1) * 1e3))
    a = np.median(np.array(rows), axis=0)
    print("K=%-5d host call %.1f us   wall to idle %.1f us (%.2f us/iteration)   device (events) %.1f us" % (K, a[0], a[1], a[1] / K, a[2]))
