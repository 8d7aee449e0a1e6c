"""round 5: the uniform-table reads of the generated body requested BSVI_SPEC_PREFETCH records ahead — iteration time of the
long loop of the README AR model (BASELINE config 1) and equality of the loss curves with the reads left where the compiler
puts them.  usage: python tools/r5/prefetch_check.py [pairs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from brancher_amd import engine, workloads as W  # noqa: E402

os.environ["BSVI_SPEC_PAIRS"] = sys.argv[1] if len(sys.argv) > 1 else "0"
ref = {}
for d in os.environ.get("SWEEP", "0 1 2 3 4").split():
    os.environ["BSVI_SPEC_PREFETCH"] = d
    for n in (64, 128, 300, 512):
        for opt in ("SGD", "Adam"):
            c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
            losses, _ = c.train(300, n, opt, seed=5, lr=1e-3)
            torch.cuda.synchronize()
            curve = losses.cpu().numpy().copy()
            key = (n, opt)
            same = "reference" if key not in ref else ("identical" if np.array_equal(curve, ref[key]) else "max diff %.3g" % np.abs(curve - ref[key]).max())
            ref.setdefault(key, curve)
            c.train(2000, n, opt, seed=0, lr=1e-3)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                c.train(20000, n, opt, seed=0, lr=1e-3)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 20000 * 1e6)
            print("prefetch %s  n %4d %-4s  %.3f us per iteration  threads %d lanes x%d  curve %s" % (
                d, n, opt, best, c.native.engine(n, 2)["n_threads"], c.native.engine(n, 2)["samples_per_lane"], same), flush=True)
