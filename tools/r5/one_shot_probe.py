"""round 5: why bench.py's timed 20-iteration call of cfg 1 (one shot behind the spin-up) takes ~125 us when the same call in a tight
loop takes ~105 us.  One-shot calls behind a spin-up of 200-iteration calls (bench.py until now), behind a spin-up of 20-iteration calls,
and with a torch event pair created in between.  usage: python tools/r5/one_shot_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from brancher_amd import engine, workloads as W  # noqa: E402

c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
run = lambda k: c.train(k, 300, "SGD", seed=0, lr=1e-3)
run(5)
run(20)
torch.cuda.synchronize()


def one_shot(chunk, spin_ms, events, reps=25, settle=0):
    wall, dev, launch, rec = [], [], [], []
    for _ in range(reps):
        t_end = time.perf_counter() + spin_ms * 1e-3
        while time.perf_counter() < t_end:
            run(chunk)
            torch.cuda.synchronize()
        for _ in range(settle):
            run(20)
            torch.cuda.synchronize()
        torch.cuda.synchronize()
        if events:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if events == "warm":                      # the HIP events exist before the region (torch creates them at the first record)
                a.record(); b.record()
            torch.cuda.synchronize()
            a.record()
        t0 = time.perf_counter()
        run(20)
        launch.append(time.perf_counter() - t0)
        if events:
            b.record()
        t_rec = time.perf_counter()
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
        rec.append(t_rec - t0 - launch[-1])
        if events:
            dev.append(a.elapsed_time(b))
    print("   first five shots in order: " + ", ".join("%.1f (call %.1f)" % (w * 1e6, l * 1e6) for w, l in list(zip(wall, launch))[:5]))
    wall.sort(); dev.sort(); launch.sort(); rec.sort()
    print("   (the library call returns after %.1f us; the closing record takes %.1f us)" % (launch[len(launch) // 2] * 1e6, rec[len(rec) // 2] * 1e6))
    print("spin-up of %3d-iteration calls for %3d ms, %d settling calls, events %-5s: one-shot wall median %.1f us (best %.1f, worst %.1f)%s"
          % (chunk, spin_ms, settle, events, wall[len(wall) // 2] * 1e6, wall[0] * 1e6, wall[-1] * 1e6,
             "; events median %.1f us" % (dev[len(dev) // 2] * 1e3) if dev else ""), flush=True)


import gc
what = sys.argv[1] if len(sys.argv) > 1 else ""
if what == "nogc":
    gc.collect()
    gc.disable()
elif what == "collect":
    gc.collect()
elif what == "disable":
    gc.disable()
elif what == "threshold":
    gc.collect()
    gc.set_threshold(1000000)
elif what == "freeze":
    gc.collect()
    gc.freeze()
print("gc:", what or "default", gc.isenabled(), gc.get_threshold(), gc.get_count())
for chunk, spin_ms, events, settle in ((200, 100, True, 0), (200, 100, "warm", 0), (200, 100, False, 0), (200, 100, True, 0), (200, 100, "warm", 0)):
    one_shot(chunk, spin_ms, events, settle=settle)
