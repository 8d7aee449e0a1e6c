"""Both engines on the same in-kernel (Philox) draws: loss, gradients, per-sample values, samples and noise of one
evaluation of a workload, specialised kernel against interpreter.  usage: python tools/jit_compare.py cfg2 [n] [offset]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brancher_amd import engine, workloads as W   # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
api = W.native_api()
build = {"cfg1": lambda: (W.build_readme_ar(api, T=20), 300), "cfg3": lambda: (W.build_readme_ar(api, T=200), 1024),
         "cfg2": lambda: (W.build_beta_binomial(api), 4096)}[which]
n_arg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
offset = int(sys.argv[3]) if len(sys.argv) > 3 else 7


def run(jit, diag):
    os.environ["BSVI_JIT"] = "1" if jit else "0"
    model, n = build()
    n = n_arg or n
    c = engine.compile_model(model, None, "pathwise")
    kw = dict(want_samples=True, want_noise=True, want_fvalues=True) if diag else {}
    res = c.evaluate(n, seed=3, offset=offset, **kw)
    out = dict(loss=float(res["loss"].item()), grads=res["grads"].cpu().numpy().copy(), bad=float(res["nonfinite_count"].item()))
    for k in ("samples", "noise", "f"):
        if k in res:
            out[k] = res[k].cpu().numpy().copy()
    return out


for diag in (True, False):
    a, b = run(True, diag), run(False, diag)
    print("diag" if diag else "lean", "loss jit %.6f interp %.6f | nonfinite %s %s | grads jit %s interp %s"
          % (a["loss"], b["loss"], a["bad"], b["bad"], a["grads"][:4], b["grads"][:4]))
    for k in ("samples", "noise", "f"):
        if k in a:
            d = np.abs(a[k] - b[k])
            print("   %s: max abs diff %.3e at %s; jit[0,:4]=%s interp[0,:4]=%s" % (k, d.max(), np.unravel_index(d.argmax(), d.shape),
                                                                                   a[k].reshape(a[k].shape[0] if a[k].ndim > 1 else 1, -1)[0, :4],
                                                                                   b[k].reshape(b[k].shape[0] if b[k].ndim > 1 else 1, -1)[0, :4]))
