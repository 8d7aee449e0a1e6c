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
def test_amortised_fixture_regenerates_bit_for_bit(tmp_path):
    regenerate("gen_golden_vae.py", ["vae_P12_H8_H6_DS20_B5_N3"], tmp_path)
    assert_same_arrays("vae_P12_H8_H6_DS20_B5_N3", tmp_path)


def test_every_fixture_was_written_by_the_deterministic_generator():
    """(`posterior_order`: the reference walks a set of variables hashed by address; the generator sorts it by name, and
    fixtures written before that cannot be regenerated)"""
    for name in golden_cases() + golden_cases(vae=True):
        meta = json.loads(str(np.load(os.path.join(GOLDEN, name + ".npz"))["meta"]))
        assert meta.get("posterior_order") == "sorted by name", name
