"""round 5: the draw service (five sample waves + three draw waves, spec_main.h) against the plain loop — long-loop iteration time of the
README AR model (BASELINE config 1) at sample counts around 300, curves compared bit for bit.  usage: python tools/r5/draw_service_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from brancher_amd import engine, workloads as W  # noqa: E402

for n in (300, 320, 256, 192, 129, 128, 64):
    ref = {}
    for service in ("0", "1"):
        os.environ["BSVI_SPEC_DRAW_SERVICE"] = service
        for opt in ("SGD", "Adam"):
            c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
            losses, _ = c.train(200, n, opt, seed=5, lr=1e-3)
            torch.cuda.synchronize()
            curve = losses.cpu().numpy().copy()
            key = (n, opt)
            if service == "0":
                ref[key] = curve
                same = "reference"
            else:
                same = "identical" if np.array_equal(curve, ref[key]) else "max diff %.3g" % np.abs(curve - ref[key]).max()
            c.train(2000, n, opt, seed=0, lr=1e-3)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                c.train(20000, n, opt, seed=0, lr=1e-3)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 20000 * 1e6)
            print("n %4d service %s %-4s  %.3f us per iteration, threads %d, curve %s" % (n, service, opt, best, c.native.engine(n, 2)["n_threads"], same), flush=True)
