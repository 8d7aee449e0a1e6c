"""round 5: what cfg 3's iteration (README AR, T = 200, 1 024 samples) costs when the standard normals come from memory instead of
being drawn (twice) inside the sweeps — the caller-supplied-noise path of the same kernels.  usage: python tools/r5/cfg3_given_noise_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from brancher_amd import engine, workloads as W  # noqa: E402

n, K = 1024, 200
c = engine.compile_model(W.build_readme_ar(W.native_api(), T=200), None, "pathwise")
for label, kw in (("Philox inside the kernel", dict()), ("normals from memory", dict(noise_seq=True))):
    if "noise_seq" in kw:
        rng = np.random.RandomState(0)
        kw = dict(noise_seq=[rng.randn(c.program.n_noise, n).astype(np.float32) for _ in range(K)])
    c.train(K, n, "SGD", seed=0, lr=1e-4, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        c.train(K, n, "SGD", seed=0, lr=1e-4, **kw)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e6)
    print("%-26s %.1f us per iteration (mode %s)" % (label, best, c.last_mode))
