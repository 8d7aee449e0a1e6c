"""round 5: a 20-iteration training call of cfg 1 on the step sequence the ranks of a multi-GPU run take (kernel, all-reduce, finalize —
one rank here, so the all-reduce is a no-op): the kept graph replayed, the graph captured anew for every call (what every call did
until round 5), and launch by launch (BSVI_GRAPH=0).  usage: python tools/r5/sharded_call_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from brancher_amd import engine, workloads as W  # noqa: E402

c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
run = lambda: c.train(20, 300, "SGD", seed=0, lr=1e-3, _force_sharded_path=True)


def measure(label, before=None, reps=40):
    times = []
    for _ in range(reps):
        if before:
            before()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    times.sort()
    print("%-44s median %8.1f us per 20-iteration call = %7.0f it/s (best %.1f us); mode %s"
          % (label, times[len(times) // 2] * 1e6, 20 / times[len(times) // 2], times[0] * 1e6, c.last_mode), flush=True)


run(); run()
measure("kept graph, replayed")
measure("graph captured anew for every call", before=lambda: c._graph_cache.clear())
os.environ["BSVI_GRAPH"] = "0"
measure("launch by launch (BSVI_GRAPH=0)")
