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
        mb = {k: [int(i) for i in v] for k, v in self.group("minibatch/").items()}
        return mb or None

    def trajectory_minibatch(self):
        seq = self.group("traj/minibatch/")
        if not seq:
            return None
        return [{k: [int(i) for i in v[it]] for k, v in seq.items()} for it in range(self.meta["trajectory"]["iters"])]

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
