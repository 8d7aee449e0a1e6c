"""round 5: two samples per lane (SPEC_PAIRS) against the one-sample kernel — bit-equality of losses / gradients / trajectories
for the README AR model (BASELINE config 1) at several shard sizes, then the long-loop iteration time of both.
usage: python tools/r5/pairs_check.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from brancher_amd import engine, workloads as W  # noqa: E402


def run(pairs, n, est, opt, steps=40):
    os.environ["BSVI_SPEC_PAIRS"] = pairs
    c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, est)
    kind = c.native.engine(n, 2)
    res = c.evaluate(n, seed=11, offset=3)
    torch.cuda.synchronize()
    ev = (float(res["loss"]), res["grads"].cpu().numpy().copy())
    losses, finite = c.train(steps, n, opt, seed=5, lr=1e-3)
    torch.cuda.synchronize()
    return kind, ev, losses.cpu().numpy().copy(), c.params.cpu().numpy().copy()


def main():
    bad = 0
    for n in (300, 130, 257, 301, 384, 512, 64, 500):
        for est in ("pathwise", "blackbox"):
            for opt in ("SGD", "Adam"):
                ka, ea, la, pa = run("0", n, est, opt)
                kb, eb, lb, pb = run("1", n, est, opt)
                same = ea[0] == eb[0] and np.array_equal(ea[1], eb[1]) and np.array_equal(la, lb) and np.array_equal(pa, pb)
                print("n %4d %-8s %-4s lanes %d/%d threads %d/%d  %s  loss %.6f  dloss %.3g dgrad %.3g dcurve %.3g" % (
                    n, est, opt, ka.get("samples_per_lane", 0), kb.get("samples_per_lane", 0), ka["n_threads"], kb["n_threads"],
                    "identical" if same else "DIFFERENT", ea[0], abs(ea[0] - eb[0]), np.abs(ea[1] - eb[1]).max(), np.abs(la - lb).max()))
                bad += 0 if same else 1
    print("mismatches:", bad)
    # timing: long loops
    for n in (300, 384, 512, 256):
        for pairs in ("0", "1"):
            os.environ["BSVI_SPEC_PAIRS"] = pairs
            c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
            c.train(2000, n, "SGD", seed=0, lr=1e-3)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                c.train(20000, n, "SGD", seed=0, lr=1e-3)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 20000 * 1e6)
            print("n %4d pairs %s: %.3f us per iteration (20 000-iteration launch), %s" % (n, pairs, best, c.native.engine(n, 2)))
    os.environ.pop("BSVI_SPEC_PAIRS", None)
    c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
    print("default at 300:", c.native.engine(300, 2))


if __name__ == "__main__":
    main()
