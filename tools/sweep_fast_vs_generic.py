"""Diagnostic: the lean build of the kernel (pre-resolved handlers) against the diagnostic build (generic
handlers, parity-tested against the reference fixtures) on the same emitted noise, over shard sizes and
chain lengths — catches layout / geometry mistakes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brancher_amd import engine, workloads as W
api = W.native_api()
worst = 0.0
for T in (5, 20, 50, 200):
    for n in (1, 63, 64, 65, 300, 1000, 1024, 5000):
        if T == 200 and n > 1024:
            continue
        c = engine.compile_model(W.build_readme_ar(api, T=T), None, "pathwise")
        a = c.evaluate(n, seed=3, offset=7, want_noise=True)
        la, ga = float(a["loss"].item()), a["grads"].clone()
        noise = a["noise"].cpu().numpy()
        named = {name: noise[s.base:s.base + s.size].T.reshape((n,) + tuple(s.shape)) for name, s in c.program.slot_by_name.items()}
        b = c.evaluate(n, noise=named)
        lb, gb = float(b["loss"].item()), b["grads"]
        p = c.evaluate(n, seed=3, offset=7)            # lean build with in-register Philox
        lp_, gp = float(p["loss"].item()), p["grads"]
        gs = float(ga.abs().max())
        e1 = max(abs(la - lb) / abs(la), float((ga - gb).abs().max()) / gs)
        e2 = max(abs(la - lp_) / abs(la), float((ga - gp).abs().max()) / gs)
        worst = max(worst, e1, e2)
        print("T=%3d N=%5d %-14s loss %.6f  rel.err given-noise %.1e  philox %.1e" % (T, n, c.native.geometry(n)["storage"], la, e1, e2))
print("worst", worst)
assert worst < 2e-5
