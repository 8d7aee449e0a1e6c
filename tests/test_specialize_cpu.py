"""Host side of the program specialiser (brancher_amd/csrc/specialize.cpp): the library turns a lowered program into a
HIP translation unit and compiles it for gfx950 with hiprtc — neither step needs a GPU, so both are checked here for
every scalar golden workload and for the BASELINE configurations.  What the generated kernels compute is checked on
the GPU (tests/test_gpu_parity.py runs every fixture through the specialised kernels AND the interpreter)."""
import re

import pytest

from conftest import Golden, golden_cases
from brancher_amd import lowering, native, workloads as W

SCALAR = [c for c in golden_cases() if not c.startswith("logreg") and not c.startswith("bnn")]      # (dense-link and Bayesian-neural-network families: their own kernels)


def lowered(case, estimator="pathwise"):
    model = Golden(case).build()
    return lowering.lower(model, model.posterior_model, estimator)


@pytest.mark.parametrize("case", SCALAR)
@pytest.mark.parametrize("estimator", ["pathwise", "blackbox"])
def test_generated_kernels_compile_for_gfx950(case, estimator):
    if estimator == "blackbox" and "T200" in case:
        pytest.skip("minutes of hiprtc on a CPU-only box (the score term keeps every model term in the forward sweep: "
                    "long live ranges); compiled and run by the GPU suite")
    program = lowered(case, estimator)
    for variant in (0, 1):
        src = native.specialised_source(program, variant)
        assert src is not None, native.load().bsvi_last_error()
        assert "#define SPEC_ESTIMATOR %d" % lowering.EST[estimator] in src
        assert native.jit_compile(src) > 0


def test_source_structure_config1():
    """README AR T=20 (BASELINE config 1): 63 instructions -> one straight-line body; every parameter-sourced uniform
    entry leaves through exactly one SPEC_DU position; the lean variant never touches a noise tensor."""
    model = W.build_readme_ar(W.native_api(), T=20)
    program = lowering.lower(model, model.posterior_model, "pathwise")
    lean, diag = native.specialised_source(program, 0), native.specialised_source(program, 1)
    positions = [int(m) for m in re.findall(r"SPEC_DU\((\d+)u,", lean)]
    assert sorted(positions) == list(range(program.n_uniform_grad))
    assert positions == [int(m) for m in re.findall(r"SPEC_DU\((\d+)u,", diag)]      # same completion order: one CSR table
    assert "noise[" not in lean and "noise[" in diag
    assert lean.count("spec_normals4(") == (program.n_noise + 3) // 4                  # one Philox call per 4 rows
    assert lean.count("spec_naff_sink(") == 41                                          # the model's log-prob terms
    assert "for (" not in lean.split("void spec_body")[1].split('#include "spec_main.h"')[0]  # no loops: fully unrolled
    assert "#define SPEC_TILE 1" in lean                                                # <= 64 positions: transpose tile
    many = native.specialised_source(program, 2)                                        # the many-workgroup geometry
    assert "#define SPEC_MAX_THREADS 256" in many and "#define SPEC_ACCUMULATE_CHUNKS 1" in many


def test_long_chain_keeps_fewer_registers():
    """T=200 (BASELINE config 3): 201 noise rows are not kept for the reverse sweep (regenerated group by group through an
    opaque group number), the launch bounds drop to 256 threads so that a lane may use the whole register file, every
    model term is deferred to the reverse step that consumes its adjoints, and contributions leave through DPP row sums"""
    model = W.build_readme_ar(W.native_api(), T=200)
    program = lowering.lower(model, model.posterior_model, "pathwise")
    src = native.specialised_source(program, 0)
    assert "#define SPEC_MAX_THREADS 256" in src and "#define SPEC_TILE 0" in src
    assert src.count("spec_normals4(") == 2 * ((program.n_noise + 3) // 4)
    assert src.count("spec_opaque(") == (program.n_noise + 3) // 4
    body = src.split("void spec_body")[1]
    turn = body.index("const float fweight")
    assert body[:turn].count("spec_naff_sink(") == 0 and body[turn:].count("spec_naff_sink(") == 401


def test_long_blackbox_program_runs_its_sinks_twice_and_compiles_without_spilling(tmp_path, monkeypatch):
    """BlackBox at T = 200 (~1 min of hiprtc, ~2 000 spilled registers).  The score term weights the
    reverse steps with the complete f, so the model terms run TWICE: value only in the forward sweep, adjoints only — at the
    deferred positions of the Pathwise schedule — in the reverse sweep; and the translation unit asks for no SLP
    vectorization (the vectorizer paired terms of distant records).  The code object must keep (nearly) every value in
    registers; the compile time follows (9 s in the build container against 52)."""
    import os
    import subprocess
    import time
    model = W.build_readme_ar(W.native_api(), T=200)
    program = lowering.lower(model, model.posterior_model, "blackbox")
    src = native.specialised_source(program, 0)
    body = src.split("void spec_body")[1]
    turn = body.index("const float fweight")
    # forward sweep: 401 values and no sink adjoints; reverse sweep: the 401 sinks again, their values going nowhere
    assert body[:turn].count("spec_naff_sink(") == 0 and body[:turn].count("T.f += 1.0f * spec_naff_lp(") == 401
    assert body[turn:].count("spec_naff_sink(") == 401 and body[turn:].count(", f_dead, gl, gs)") == 401
    assert "// bsvi-jit-option: -fno-slp-vectorize" in src
    pathwise = native.specialised_source(lowering.lower(model, model.posterior_model, "pathwise"), 0)
    assert "bsvi-jit-option" not in pathwise and "f_dead" not in pathwise
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("no llvm-readelf in this image")
    dump = str(tmp_path / "kernel.co")
    monkeypatch.setenv("BSVI_JIT_DUMP", dump)
    t0 = time.perf_counter()
    assert native.jit_compile(src + "\n// (unique: not served from the code cache)\n") > 0
    seconds = time.perf_counter() - t0
    notes = subprocess.run([readelf, "--notes", dump], capture_output=True, text=True).stdout
    meta = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_spill_count|vgpr_count):\s+(\d+)", notes)}
    assert meta["vgpr_spill_count"] <= 64, meta            # (12 here; 2 158 before)
    assert seconds < 40.0, seconds                         # (9 s here; 52 s before — a loose bound: shared CI cores)


def test_unrolling_limit_declines():
    """a program whose unrolled stream exceeds the generator's limit is left to the interpreter (not an error)"""
    api = W.native_api()
    model = W.build_multivariate_regression(api, n=100) if hasattr(W, "build_multivariate_regression") else None
    if model is None:
        pytest.skip("no wide workload builder")
    program = lowering.lower(model, model.posterior_model, "pathwise")
    src = native.specialised_source(program, 0)
    # either outcome is legal; when declined the reason is reported
    if src is None:
        assert b"instruction visits" in native.load().bsvi_last_error()


def test_config1_kernel_has_no_spills(tmp_path, monkeypatch):
    """the training-loop kernel of BASELINE config 1: every per-sample value in a register — no scratch memory, no spilled
    registers (BSVI_JIT_DUMP writes the code object hiprtc produced; llvm-readelf --notes reads its metadata)"""
    import os
    import subprocess
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("no llvm-readelf in this image")
    model = W.build_readme_ar(W.native_api(), T=20)
    program = lowering.lower(model, model.posterior_model, "pathwise")
    dump = str(tmp_path / "kernel.co")
    monkeypatch.setenv("BSVI_JIT_DUMP", dump)
    assert native.jit_compile(native.specialised_source(program, 0) + "\n// (unique: not served from the code cache)\n") > 0
    notes = subprocess.run([readelf, "--notes", dump], capture_output=True, text=True).stdout
    meta = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count):\s+(\d+)", notes)}
    assert meta["private_segment_fixed_size"] == 0 and meta["vgpr_spill_count"] == 0, meta
    assert meta["vgpr_count"] <= 256                      # up to 8 waves (512 threads) of one workgroup: 256 registers per lane
    # (scalar registers do spill — ~40 words to vector-register lanes, v_writelane / v_readlane: the 20 round keys of the Philox
    #  schedule shared by the calls of a draw, the optimizer's constants.  Forcing the keys to be recomputed per call removes a
    #  third of them but serialises the calls: 190 -> 163 k it/s)
    assert meta["sgpr_spill_count"] <= 64, meta


def test_code_object_cache_on_disk(tmp_path):
    """A second PROCESS does not pay hiprtc again: the code object of a generated translation unit is stored under the
    cache directory, keyed by source + embedded headers + options + toolchain version.  Each step below runs in a fresh
    interpreter (the in-process cache would hide the disk one)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import json, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "from brancher_amd import lowering, native, workloads as W\n"
        "m = W.build_readme_ar(W.native_api(), T=20)\n"
        "src = native.specialised_source(lowering.lower(m, m.posterior_model, 'pathwise'), 0)\n"
        "native.load()\n"
        "t0 = time.perf_counter(); n, origin = native.jit_load(src); t1 = time.perf_counter()\n"
        "n2, origin2 = native.jit_load(src)\n"
        "print(json.dumps(dict(n=n, origin=origin, again=origin2, ms=(t1 - t0) * 1e3, dir=native.jit_cache_dir())))\n" % root)

    def run(**env):
        out = subprocess.run([sys.executable, "-c", script], check=True, capture_output=True, text=True,
                             env=dict(os.environ, BSVI_CACHE_DIR=str(tmp_path), **env)).stdout
        return json.loads(out.strip().splitlines()[-1])

    first = run()
    assert first["origin"] == "hiprtc" and first["again"] == "process cache" and first["dir"] == str(tmp_path)
    files = [f for f in os.listdir(str(tmp_path)) if f.endswith(".co")]
    assert len(files) == 1 and not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]
    second = run()
    assert second["origin"] == "disk cache" and second["n"] == first["n"]
    assert second["ms"] < 50.0, second          # a second process's cold start of cfg 1 under 50 ms
    # a damaged file is not trusted: it is recompiled and replaced
    path = os.path.join(str(tmp_path), files[0])
    blob = bytearray(open(path, "rb").read())
    blob[len(blob) // 2] ^= 0xFF
    open(path, "wb").write(bytes(blob))
    third = run()
    assert third["origin"] == "hiprtc" and third["n"] == first["n"]
    assert run()["origin"] == "disk cache"
    # switched off: no directory, no file read
    off = run(BSVI_JIT_CACHE="0")
    assert off["origin"] == "hiprtc" and off["dir"] == ""


def _jit_child(script_body, env):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = ("import hashlib, json, os, sys\nsys.path.insert(0, %r)\n"
              "from brancher_amd import lowering, native, workloads as W\napi = W.native_api()\n" % root) + script_body
    out = subprocess.run([sys.executable, "-c", script], check=True, capture_output=True, text=True, env=dict(os.environ, **env)).stdout
    return json.loads(out.strip().splitlines()[-1])


def test_a_code_object_is_a_function_of_what_the_cache_key_hashes(tmp_path):
    """VERDICT r4 item 6.  The same generated translation unit compiled as the FIRST compilation of a process and after a dozen
    other programs of the same process (one of them with the per-program -fno-slp-vectorize option) gives byte-identical code
    objects: hiprtc keeps no state between compilations that reaches the code.  (tools/r5/jit_determinism.py runs the long form.)"""
    body = (
        "def srcs(b, e, **kw):\n"
        "    m = getattr(W, b)(api, **kw)\n"
        "    p = lowering.lower(m, m.posterior_model, e)\n"
        "    return [native.specialised_source(p, v) for v in range(2)]\n"
        "if os.environ['N_OTHER'] != '0':\n"
        "    for b, e, kw in [('build_readme_ar', 'pathwise', dict(T=3)), ('build_readme_ar', 'blackbox', dict(T=4)),\n"
        "                     ('build_beta_binomial', 'pathwise', {}), ('build_lognormal_normal', 'blackbox', {})]:\n"
        "        for s in srcs(b, e, **kw):\n"
        "            native.jit_compile(('// bsvi-jit-option: -fno-slp-vectorize\\n' if e == 'blackbox' else '') + s)\n"
        "res = {}\n"
        "for v, s in enumerate(srcs('build_readme_ar', 'pathwise', T=20)):\n"
        "    os.environ['BSVI_JIT_DUMP'] = os.path.join(os.environ['OUT'], 'v%d.co' % v)\n"
        "    native.jit_compile(s)\n"
        "    res[str(v)] = hashlib.sha256(open(os.environ['BSVI_JIT_DUMP'], 'rb').read()).hexdigest()\n"
        "print(json.dumps(res))\n")
    first = _jit_child(body, dict(N_OTHER="0", OUT=str(tmp_path), BSVI_JIT_CACHE="0"))
    late = _jit_child(body, dict(N_OTHER="8", OUT=str(tmp_path), BSVI_JIT_CACHE="0"))
    assert first == late


def test_the_cache_key_carries_the_identity_of_the_compiler(tmp_path):
    """The root cause of round 4's one-ulp failures (profiles/r5/jit_cache_root_cause.txt): which libamd_comgr — i.e. which clang —
    compiles a translation unit depends on what the process loaded first (the ROCm libraries bundled with torch, or the system's
    when rocprofv3 or LD_PRELOAD brings them in), the two compilers produce different code for the same source, and the key did
    not say which one had run.  Now a process with another compiler misses the entries of the first, stores its own, and each
    context is served its own code afterwards."""
    import os
    system_comgr = "/opt/rocm/lib/libamd_comgr.so.3"
    if not os.path.exists(system_comgr):
        pytest.skip("no system libamd_comgr to bring in")
    body = (
        "m = W.build_readme_ar(api, T=5)\n"
        "src = native.specialised_source(lowering.lower(m, m.posterior_model, 'pathwise'), 0)\n"
        "n, origin = native.jit_load(src)\n"
        "print(json.dumps(dict(origin=origin, n=n, identity=native.jit_compiler_identity())))\n")
    env = dict(BSVI_CACHE_DIR=str(tmp_path))
    preload = dict(env, LD_PRELOAD=":".join([system_comgr] + [p for p in os.environ.get("LD_PRELOAD", "").split(":") if p]))
    a = _jit_child(body, env)
    b = _jit_child(body, preload)
    if a["identity"] == b["identity"]:
        pytest.skip("the process already runs the system's compiler")
    assert a["origin"] == "hiprtc" and b["origin"] == "hiprtc", (a, b)       # (round 4: the second would have been "disk cache")
    assert len([f for f in os.listdir(str(tmp_path)) if f.endswith(".co")]) == 2
    assert _jit_child(body, env)["origin"] == "disk cache" and _jit_child(body, preload)["origin"] == "disk cache"
