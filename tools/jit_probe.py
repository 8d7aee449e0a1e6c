"""Specialised kernels against the interpreter on one GPU: same seeds, both engines; then timing of the training loop.
usage: python tools/jit_probe.py [cfg1|cfg2|cfg3] [iterations]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brancher_amd import engine, workloads as W   # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
api = W.native_api()
build = {"cfg1": lambda: (W.build_readme_ar(api, T=20), 300), "cfg3": lambda: (W.build_readme_ar(api, T=200), 1024),
         "cfg2": lambda: (W.build_beta_binomial(api), 4096)}[which]


def run(jit):
    os.environ["BSVI_JIT"] = "1" if jit else "0"
    model, n = build()
    c = engine.compile_model(model, None, "pathwise")
    info = c.native.engine(n, 2), c.native.engine(n, 0)
    res = c.evaluate(n, seed=3, offset=7)
    loss, grads = float(res["loss"].item()), res["grads"].cpu().numpy().copy()
    t0 = time.time()
    c.evaluate(n, seed=3, offset=8)
    torch.cuda.synchronize()
    first = time.time() - t0
    out = {}
    for opt, kw in (("SGD", dict(lr=1e-3)), ("Adam", dict(lr=1e-2))):
        losses, finite = c.train(50, n, opt, seed=5, **kw)       # warm-up (and hiprtc)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        losses, finite = c.train(iters, n, opt, seed=5, **kw)
        ev1.record()
        torch.cuda.synchronize()
        out[opt] = (ev0.elapsed_time(ev1) * 1e3 / iters, float(losses[-1].item()), c.last_mode)
    return info, loss, grads, out


ia, la, ga, ta = run(True)
ib, lb, gb, tb = run(False)
print("engines:", ia, "|", ib)
print("loss jit %.6f interp %.6f  rel %.2e" % (la, lb, abs(la - lb) / abs(lb)))
print("grad max rel diff %.2e" % (np.abs(ga - gb).max() / np.abs(gb).max()))
for opt in ta:
    print("%s: jit %.2f us/it (%s, final loss %.4f) | interpreter %.2f us/it (%s, final loss %.4f)" % ((opt,) + ta[opt][0:1] + (ta[opt][2], ta[opt][1]) + tb[opt][0:1] + (tb[opt][2], tb[opt][1])))
