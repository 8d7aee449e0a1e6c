"""The committed fixtures can be audited: regenerating a case with the committed generator (oracle/gen_golden*.py, which
imports the REAL reference from /root/reference) gives the same arrays.  Container-only — the reference does not travel to
the GPU box, so the test skips where it is absent."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_cases

REFERENCE = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "brancher")),
                                     reason="the reference is only present in the build container")


def regenerate(script, cases, out_dir):
    env = dict(os.environ, BSVI_GOLDEN_OUT=str(out_dir), PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="1")
    subprocess.run([sys.executable, os.path.join(ROOT, "oracle", script)] + list(cases), check=True, env=env,
                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=900)


def assert_same_arrays(name, out_dir):
    new, old = np.load(os.path.join(str(out_dir), name + ".npz")), np.load(os.path.join(GOLDEN, name + ".npz"))
    assert sorted(new.files) == sorted(old.files)
    for key in old.files:
        if key == "meta":
            assert json.loads(str(new[key])) == json.loads(str(old[key]))
        else:
            assert np.array_equal(new[key], old[key], equal_nan=True), (name, key)


@needs_reference
def test_scalar_and_dense_fixtures_regenerate_bit_for_bit(tmp_path):
    cases = ["readme_ar_T5_N7", "beta_binomial_N512", "logreg_C3_P6_DS20_B12_N5", "flat_vector_sum_d5_N48"]
    regenerate("gen_golden.py", cases, tmp_path)
    for name in cases:
        assert_same_arrays(name, tmp_path)


@needs_reference
def test_the_headline_fixture_regenerates(tmp_path):
    """`readme_ar_T20_N300` — BASELINE config 1, the fixture the headline's parity rests on (and the one inside
    tests/c_abi/readme_ar_T20_N300.blob).  The draws and the three built-in estimators' losses regenerate bit for bit;
    every other array — per-sample terms, gradients, the records of the two user-defined estimators, the trajectory — to
    rounding, 2e-6 of the array's scale (gradients: of their record's largest entry): torch's reductions over 300 samples choose their summation order by how the
    reference's graph happens to sit in memory (sums over a `set` of variables hashed by address feed them), which a
    second process does not repeat.  Five times below the 1e-5 the parity tests hold the kernels to."""
    name = "readme_ar_T20_N300"
    regenerate("gen_golden.py", [name], tmp_path)
    new, old = np.load(os.path.join(str(tmp_path), name + ".npz")), np.load(os.path.join(GOLDEN, name + ".npz"))
    assert sorted(new.files) == sorted(old.files)
    exact_keys = [k for k in old.files if k.startswith("noise/") or k in ("loss_pathwise", "loss_blackbox", "loss_taylor1")]
    assert "loss_pathwise" in exact_keys and "loss_blackbox" in exact_keys and any(k.startswith("noise/") for k in exact_keys)
    worst = {}
    for key in old.files:
        if key == "meta":
            assert json.loads(str(new[key])) == json.loads(str(old[key]))
        elif key in exact_keys:
            assert np.array_equal(new[key], old[key], equal_nan=True), key
        else:
            # (gradients: against the largest entry of their estimator's record, the scale every parity check uses)
            group = key.split("/")[0] + "/" if key.startswith("grad_") else key
            scale = max(max(float(np.abs(old[k]).max()) for k in old.files if k.startswith(group)), 1e-30)
            worst[group] = max(worst.get(group, 0.0), float(np.abs(new[key].astype(np.float64) - old[key]).max()) / scale)
    # (observed in this container: 1.2e-6 for `loss_custom_baseline`, 3e-7 for its gradients, 1.5e-7 for the BlackBox gradients)
    for group, w in worst.items():
        assert w <= 2e-6, (group, w)


@needs_reference
def test_amortised_fixture_regenerates_bit_for_bit(tmp_path):
    regenerate("gen_golden_vae.py", ["vae_P12_H8_H6_DS20_B5_N3"], tmp_path)
    assert_same_arrays("vae_P12_H8_H6_DS20_B5_N3", tmp_path)


def test_every_fixture_was_written_by_the_deterministic_generator():
    """(`posterior_order`: the reference walks a set of variables hashed by address; the generator sorts it by name, and
    fixtures written before that cannot be regenerated)"""
    for name in golden_cases() + golden_cases(vae=True):
        meta = json.loads(str(np.load(os.path.join(GOLDEN, name + ".npz"))["meta"]))
        assert meta.get("posterior_order") == "sorted by name", name
