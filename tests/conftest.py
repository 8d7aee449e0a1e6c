import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# (Round 4 gave the suite a code-object cache directory of its own because bit-equality tests between kernel variants had failed by
#  one ulp after a profiling script.  Root cause, round 5: code objects compiled under rocprofv3 — by the SYSTEM's clang, which the
#  profiler's tool library loads ahead of the one bundled with torch — were served under the same key to unprofiled processes.  The
#  key now carries the compiler's identity (specialize.cpp, tests/test_specialize_cpu.py), and the suite uses the user's cache
#  directory like any other process.)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases(vae=False):
    # vae_* fixtures (amortised workload, oracle/gen_golden_vae.py) have their own tests
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith("vae_") == vae)


class Golden:
    """One fixture generated from the real reference by oracle/gen_golden.py."""

    def __init__(self, name):
        self.name = name
        self.data = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(str(self.data["meta"]))
        self.N = self.meta["N"]

    def group(self, prefix):
        return {k[len(prefix):]: self.data[k] for k in self.data.files if k.startswith(prefix)}

    @property
    def noise(self):
        return self.group("noise/")

    def build(self, api=None):
        from brancher_amd import workloads as W
        return getattr(W, self.meta["builder"])(api or W.native_api(), **self.meta["kwargs"])

    @property
    def minibatch(self):
        """what the reference drew for its OBSERVED variables in the recorded evaluation: the rows of every minibatch index variable
        ("minibatch/<name>") and — round 6 — the value of every variable observed by flag only ("drawn/<name>": the reference draws it
        from its own distribution once per evaluation, variables.py:553-565)"""
        mb = {k: [int(i) for i in v] for k, v in self.group("minibatch/").items()}
        mb.update({k: np.asarray(v) for k, v in self.group("drawn/").items()})
        return mb or None

    def trajectory_minibatch(self):
        seq, drawn = self.group("traj/minibatch/"), self.group("traj/drawn/")
        if not seq and not drawn:
            return None
        out = []
        for it in range(self.meta["trajectory"]["iters"]):
            step = {k: [int(i) for i in v[it]] for k, v in seq.items()}
            step.update({k: np.asarray(v[it]) for k, v in drawn.items()})
            out.append(step)
        return out

    def trajectory_noise(self):
        tr = self.meta["trajectory"]
        seq = self.group("traj/noise/")
        return [{k: v[i] for k, v in seq.items()} for i in range(tr["iters"])]

    def opt_kwargs(self):
        tr = self.meta["trajectory"]
        return {k: v for k, v in tr.items() if k not in ("iters", "n", "optimizer")}


@pytest.fixture(params=golden_cases())
def golden(request):
    return Golden(request.param)


@pytest.fixture(params=golden_cases(vae=True))
def vae_golden(request):
    return Golden(request.param)


def rel_err(a, b, floor=1e-7):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + floor))


def yardstick_grad_check(named, exact, reference, tol=1e-5):
    """The bound of every check that cannot hold the flat 1e-5 of BASELINE.json's north_star: the truth is the oracle in DOUBLE
    precision on the same draws, and the kernel must be as close to it as the reference arithmetic — single precision — is
    itself (x4), or within 1e-5 of the largest gradient:  err <= max(4 * |reference_fp32 - oracle_fp64|, 1e-5 * scale).
    (BlackBox gradients multiply log q by f, two sums of opposite sign: the reference's own single-precision result is
    1e-5 ... 1e-4 of the scale away from the double-precision one.  A flat 1e-4 said nothing about WHICH of the two is off.)"""
    scale = max(np.abs(v).max() for v in exact.values() if v is not None)
    for name, g64 in exact.items():
        g64 = np.zeros(1) if g64 is None else g64
        ref = np.zeros(1) if reference.get(name) is None else reference[name]
        err, yard = np.abs(named[name] - g64).max(), np.abs(ref - g64).max()
        assert err <= max(4 * yard, tol * scale), (name, err, yard, scale)


def exact_oracle(g_or_model, n, estimator, noise, minibatch=None):
    import torch as _t
    from oracle.svi_oracle import Oracle
    model = g_or_model.build() if isinstance(g_or_model, Golden) else g_or_model
    return Oracle(model, dtype=_t.float64).loss_and_grads(n, estimator, noise, minibatch)


def yardstick_loss_check(loss, exact_loss, reference_loss, tol=1e-5):
    """the same bound for the value: err <= max(4 * |reference_fp32 - oracle_fp64|, 1e-5 * |oracle_fp64|)"""
    err, yard = abs(loss - exact_loss), abs(reference_loss - exact_loss)
    assert err <= max(4 * yard, tol * abs(exact_loss)), (loss, exact_loss, reference_loss)
