"""The C ABI exercised by a C program (tests/c_abi/check_abi.c): compiled against include/bsvi.h and the HIP runtime API
only, it loads a committed program blob of the README autoregressive model with the noise of the reference fixture
`readme_ar_T20_N300` and compares loss and gradients with the reference's.  On the CPU: the program compiles and links
against the library, and the committed blob is what tests/c_abi/make_blob.py writes today."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "c_abi", "check_abi.c")
BLOB = os.path.join(ROOT, "tests", "c_abi", "readme_ar_T20_N300.blob")


def build(tmp_path):
    exe = str(tmp_path / "check_abi")
    lib_dir = os.path.join(ROOT, "brancher_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", SRC, "-o", exe, "-L" + lib_dir, "-lbsvi", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_program_builds_against_the_header_and_the_blob_is_current(tmp_path):
    build(tmp_path)
    before = open(BLOB, "rb").read()
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "c_abi", "make_blob.py")], stdout=subprocess.DEVNULL)
        assert open(BLOB, "rb").read() == before, "tests/c_abi/readme_ar_T20_N300.blob is stale: commit the regenerated file"
    finally:
        open(BLOB, "wb").write(before)


@pytest.mark.gpu
def test_c_program_reproduces_the_reference_through_the_c_abi(tmp_path):
    exe = build(tmp_path)
    run = subprocess.run([exe, BLOB], capture_output=True, text=True, timeout=300)
    sys.stdout.write(run.stdout)
    sys.stderr.write(run.stderr)
    assert run.returncode == 0, (run.returncode, run.stderr[-400:])
    assert "C ABI ok" in run.stdout
