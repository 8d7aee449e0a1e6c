"""The two engines of the scalar path against each other and against the reference fixtures: the program-specialised
kernels libbsvi generates and compiles with hiprtc (brancher_amd/csrc/specialize.cpp) and the
interpreter kernels of elbo_kernel.hip (BSVI_JIT=0).  tests/test_gpu_parity.py runs every fixture through the default
engine (specialised); here the interpreter is forced for the same fixtures, the engines are compared on in-kernel
(Philox) draws, and the launch geometries / training modes of the specialised kernels are compared with each other.
Also: hand-written inference loops in the reference's style through the device optimizer."""
import os

import numpy as np
import pytest
import torch

from conftest import Golden, golden_cases, rel_err, yardstick_grad_check, yardstick_loss_check, exact_oracle
from brancher_amd import engine, inference, workloads as W
from brancher_amd.optimizers import ProbabilisticOptimizer

pytestmark = pytest.mark.gpu
TOL = 1e-5
SCALAR = [c for c in golden_cases() if not c.startswith("logreg") and not c.startswith("bnn")]      # (dense-link and Bayesian-neural-network families: their own kernels)


@pytest.fixture
def interpreter(monkeypatch):
    monkeypatch.setenv("BSVI_JIT", "0")


def grad_check(named, ref, tol):
    scale = max(np.abs(g).max() for g in ref.values())
    for name, g_ref in ref.items():
        assert np.abs(named[name] - g_ref).max() <= tol * scale + 1e-7 * scale, name


@pytest.mark.skipif(os.environ.get("BSVI_JIT") == "0", reason="the suite is being run on the interpreter engine")
def test_default_engine_is_the_specialised_kernel():
    c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
    for mode in (0, 1, 2):
        info = c.native.engine(300, mode)
        # (round 5: the in-kernel loop of five sample waves runs with three draw waves that draw for all five)
        assert info["engine"] == "specialised" and info["n_blocks"] == 1 and info["n_threads"] == (512 if mode == 2 else 320)
    # up to three sample waves the in-kernel loop (mode 2) has one wave more than the samples need: it draws for the owners' wave;
    # four sample waves get four draw waves, five get three (the draw service: no sample wave draws after the first iteration)
    for n, waves, extra in ((50, 1, 1), (128, 2, 1), (192, 3, 1), (256, 4, 4), (300, 5, 3)):
        assert c.native.engine(n, 1)["n_threads"] == 64 * waves and c.native.engine(n, 2)["n_threads"] == 64 * (waves + extra)
    many = c.native.engine(262144, 1)
    assert many["engine"] == "specialised" and many["n_threads"] == 256 and many["n_blocks"] <= 512
    # (round 4: the in-kernel loop also runs over several workgroups — workgroup 0 owns the iteration)
    loop = c.native.engine(262144, 2)
    assert loop["engine"] == "specialised" and loop["n_blocks"] == many["n_blocks"]


@pytest.mark.parametrize("case", SCALAR)
@pytest.mark.parametrize("estimator", ["pathwise", "blackbox"])
def test_interpreter_matches_reference_golden(case, estimator, interpreter):
    g = Golden(case)
    c = engine.compile_model(g.build(), None, estimator)
    assert c.native.engine(g.N, 0)["engine"] == "interpreter"
    res = c.evaluate(g.N, noise=g.noise, minibatch=g.minibatch)
    ref = float(g.data["loss_" + estimator])
    if estimator == "pathwise" and case not in ("gp_hyperparameters_n32_N40", "gp_hyperparameters_n100_N24") and not case.startswith(("gp_marginal", "gp_structured", "prf_")):
        assert abs(float(res["loss"].item()) - ref) <= TOL * abs(ref)
        grad_check(c.named_grads(), g.group("grad_%s/" % estimator), TOL)
        return
    # BlackBox (log q times f: two sums of opposite sign) and the batched multivariate-normal terms (two single-precision
    # Cholesky factorisations at a condition number of ~1e3 agree to ~1e-4): the yardstick of tests/test_gpu_parity.py — as
    # close to the double-precision oracle as the reference's own single-precision record is (x4), or 1e-5 of the scale
    exact = exact_oracle(g, g.N, estimator, g.noise, g.minibatch)
    yardstick_loss_check(float(res["loss"].item()), exact["loss"], ref)
    yardstick_grad_check(c.named_grads(), exact["grads"], g.group("grad_%s/" % estimator))


@pytest.mark.parametrize("builder,kwargs,n", [
    ("build_readme_ar", dict(T=20), 300), ("build_readme_ar", dict(T=20), 5000), ("build_readme_ar", dict(T=60), 700),
    ("build_readme_ar", dict(T=200), 1024), ("build_beta_binomial", dict(n_obs=30), 4096),
    ("build_heavy_tails", dict(n_obs=12), 1000), ("build_beta_ar", dict(T=20), 640)])
@pytest.mark.parametrize("estimator", ["pathwise", "blackbox"])
def test_engines_agree_on_philox_draws(builder, kwargs, n, estimator, monkeypatch):
    """same seed, same offset: both engines draw the same samples (one Philox stream, one set of transforms) and give
    the same loss and gradients up to summation order"""
    out = {}
    for jit in ("1", "0"):
        monkeypatch.setenv("BSVI_JIT", jit)
        c = engine.compile_model(getattr(W, builder)(W.native_api(), **kwargs), None, estimator)
        res = c.evaluate(n, seed=11, offset=5, want_samples=True)
        out[jit] = (float(res["loss"].item()), res["grads"].cpu().numpy().copy(), res["samples"].cpu().numpy().copy())
    (la, ga, sa), (lb, gb, sb) = out["1"], out["0"]
    assert np.abs(sa - sb).max() <= 2e-6 * max(1.0, np.abs(sb).max())
    assert abs(la - lb) <= 2e-6 * abs(lb)
    assert np.abs(ga - gb).max() <= (2e-6 if estimator == "pathwise" else 2e-5) * np.abs(gb).max()


def test_very_long_program_compiled_with_the_basic_allocator_agrees_with_the_interpreter(monkeypatch):
    """A Gaussian process over 200 inputs with LATENT function values: 814 instructions, 604 noise rows, and a 200 x 200
    covariance per sample that the batched kernel keeps in device memory.  Past 700 instructions the generated kernel is compiled
    with LLVM's basic register allocator (specialize.cpp kBasicRegallocAboveCode: 14 s instead of 150 s) — the same arithmetic,
    so it must agree with the interpreter on the same Philox draws; and the training loop must run on it."""
    out = {}
    for jit in ("1", "0"):
        monkeypatch.setenv("BSVI_JIT", jit)
        c = engine.compile_model(W.build_gp_hyperparameters(W.native_api(), n=200, jitter=5e-2), None, "pathwise")
        assert c.native.engine(48, 0)["engine"] == ("specialised" if jit == "1" else "interpreter")
        res = c.evaluate(48, seed=7, offset=3)
        out[jit] = (float(res["loss"].item()), res["grads"].cpu().numpy().copy())
    (la, ga), (lb, gb) = out["1"], out["0"]
    assert np.isfinite(la) and np.isfinite(ga).all()
    assert abs(la - lb) <= 1e-5 * abs(lb)
    assert np.abs(ga - gb).max() <= 1e-5 * np.abs(gb).max()
    monkeypatch.setenv("BSVI_JIT", "1")
    model = W.build_gp_hyperparameters(W.native_api(), n=200, jitter=5e-2)
    from brancher_amd import inference
    inference.perform_inference(model, number_iterations=30, number_samples=32, optimizer="Adam", lr=0.02)
    losses = np.asarray(model.diagnostics["loss curve"])
    assert np.isfinite(losses).all() and losses[-10:].mean() < losses[:10].mean()


@pytest.mark.parametrize("optimizer,kw", [("SGD", dict(lr=1e-3)), ("SGD", dict(lr=1e-3, momentum=0.9, nesterov=True)),
                                          ("Adam", dict(lr=1e-2)), ("Adam", dict(lr=1e-2, amsgrad=True, weight_decay=1e-3))])
def test_in_kernel_loop_equals_launch_per_iteration(optimizer, kw):
    """the specialised kernel's training loop (one launch) against its own single-iteration launches (bsvi_svi_step) and
    the multi-GPU step sequence replayed from HIP graphs: same noise, same optimizer arithmetic"""
    curves, params = [], []
    for opts in (dict(), dict(allow_persistent=False), dict(_force_sharded_path=True)):
        c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
        losses, finite = c.train(40, 300, optimizer, seed=3, **opts, **kw)
        assert bool(finite.all())
        curves.append(losses.cpu().numpy())
        params.append(c.params.cpu().numpy().copy())
        assert c.last_mode == ("persistent", "stepwise", "graph")[len(curves) - 1]
    for other in (1, 2):
        assert rel_err(curves[other], curves[0]) <= 2e-6
        assert np.abs(params[other] - params[0]).max() <= 2e-5


@pytest.mark.parametrize("n", [50, 64, 100, 128, 200, 256, 257, 300, 320])
@pytest.mark.parametrize("optimizer,kw", [("SGD", dict(lr=1e-3)), ("Adam", dict(lr=1e-2))])
def test_draw_wave_loop_equals_plain_loop(n, optimizer, kw, monkeypatch):
    """the loop kernel with the extra wave that draws for the owners' wave (up to four sample waves), or — five sample waves, BASELINE
    config 1 — with the three draw waves that draw for all of them (the draw service), against the same loop without and against
    launch-per-iteration: the draws depend on (seed, offset, sample, row) only — bit-identical curves"""
    curves = []
    for env, opts in (("1", dict()), ("0", dict()), ("1", dict(allow_persistent=False))):
        monkeypatch.setenv("BSVI_SPEC_DRAW_WAVE", env)
        c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
        losses, finite = c.train(30, n, optimizer, seed=4, **opts, **kw)
        assert bool(finite.all())
        curves.append(losses.cpu().numpy())
    assert np.array_equal(curves[0], curves[1])
    assert rel_err(curves[2], curves[0]) <= 2e-6


@pytest.mark.parametrize("builder,kwargs,n,optimizer,kw", [
    ("build_readme_ar", dict(T=20), 1024, "SGD", dict(lr=1e-3)),            # 4 workgroups
    ("build_readme_ar", dict(T=20), 1500, "Adam", dict(lr=2e-3)),           # 6 workgroups, a ragged last one
    ("build_beta_binomial", dict(), 4096, "SGD", dict(lr=0.1)),             # BASELINE config 2: 16 workgroups
    ("build_readme_ar", dict(T=200), 1024, "SGD", dict(lr=1e-4)),           # BASELINE config 3's shard
])
def test_loop_over_several_workgroups_equals_launch_per_iteration(builder, kwargs, n, optimizer, kw, monkeypatch):
    """a shard of several workgroups trained in ONE launch (workgroup 0 owns the iteration: it adds the rows, steps, writes the
    parameters and releases the generation number the others wait on) against one launch per iteration — the rows are added in
    the same order by the same code, so under SGD the curves agree to the bit — and against BSVI_SPEC_LOOP_MANY=0"""
    curves, params, modes = [], [], []
    for env, opts in (("1", dict()), ("1", dict(allow_persistent=False)), ("0", dict())):
        monkeypatch.setenv("BSVI_SPEC_LOOP_MANY", env)
        c = engine.compile_model(getattr(W, builder)(W.native_api(), **kwargs), None, "pathwise")
        a, fa = c.train(17, n, optimizer, seed=4, **opts, **kw)
        b, fb = c.train(8, n, optimizer, seed=4, **opts, **kw)             # (a second call: the generation numbers go on)
        assert bool(fa.all()) and bool(fb.all())
        curves.append(np.concatenate([a.cpu().numpy(), b.cpu().numpy()]))
        params.append(c.params.cpu().numpy().copy())
        modes.append(c.last_mode)
    assert modes[0] == "persistent" and modes[1] == "stepwise", modes
    # (Adam's bias corrections are running products inside a launch and powers at its start: equal to rounding)
    if optimizer == "SGD":
        assert np.array_equal(curves[0], curves[1]) and np.array_equal(params[0], params[1])
    assert rel_err(curves[1], curves[0]) <= 2e-6 and rel_err(curves[2], curves[0]) <= 2e-6


def test_pretraining_iterations_in_the_loop_kernel(monkeypatch):
    """model parameters are stepped only after `pretraining_iterations` (inference.py:102-104): in-kernel loop vs the
    interpreter's persistent trainer"""
    def run(jit):
        monkeypatch.setenv("BSVI_JIT", jit)
        model = W.build_learnable_model(W.native_api())
        c = engine.compile_model(model, None, "pathwise")
        losses, _ = c.train(30, 60, "Adam", seed=5, pretraining_iterations=10, lr=5e-2)
        return losses.cpu().numpy(), c.params.cpu().numpy().copy()
    (la, pa), (lb, pb) = run("1"), run("0")
    assert rel_err(la, lb) <= 5e-6
    assert np.abs(pa - pb).max() <= 5e-5 * (1 + np.abs(pb).max())


def test_many_workgroup_geometry_sums_like_one_workgroup():
    """a shard larger than one workgroup: 256-thread workgroups walking sample chunks, rows of sums added by the last one
    to arrive — against the same samples evaluated as shards that each fit one workgroup (linearity of the sums)"""
    c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
    n = 256 * 600 + 37                       # more workgroups than the launch has -> several chunks each, ragged tail
    res = c.evaluate(n, seed=21, offset=2)
    total = c.out.cpu().numpy().astype(np.float64).copy()
    again = c.evaluate(n, seed=21, offset=2)
    assert np.array_equal(c.out.cpu().numpy().astype(np.float64), total)            # bitwise reproducible
    import ctypes as C
    from brancher_amd import native
    acc = np.zeros_like(total)
    for base in range(0, n, 512):
        m = min(512, n - base)
        args = c._elbo_args(m, n, base, None, 21, 2)
        native.check(c.lib.bsvi_elbo_fwd_bwd(c.native.handle, C.byref(args)))
        acc += c.out.cpu().numpy().astype(np.float64)
    scale = float(n)
    assert abs(acc[0] * -1.0 / scale - total[2]) <= 2e-6 * abs(total[2])
    assert np.abs(acc[4:] * -1.0 / scale - total[4:]).max() <= 5e-6 * np.abs(total[4:]).max()


class _ReferenceStyleKL(inference.InferenceMethod):
    """an InferenceMethod as a user of the reference would write it (`inference.py:114-151`): compute_loss only"""
    learnable_model, learnable_sampler, needs_sampler = True, False, False

    def check_model_compatibility(self, joint_model, posterior_model, sampler_model):
        pass

    def compute_loss(self, joint_model, posterior_model, sampler_model, number_samples, input_values={}):
        return -joint_model.estimate_log_model_evidence(number_samples=number_samples, method="ELBO", for_gradient=True,
                                                        posterior_model=posterior_model)

    def correct_gradient(self, joint_model, posterior_model, sampler_model, number_samples, input_values={}):
        pass

    def post_process(self, joint_model):
        pass


@pytest.mark.parametrize("optimizer,kw", [("SGD", dict(lr=1e-3)), ("Adam", dict(lr=1e-2))])
def test_reference_style_inference_method_drives_the_device_kernels(optimizer, kw):
    """perform_inference with a method that only implements compute_loss: the Python loop of inference.py:95-108 over the
    fused evaluation and ProbabilisticOptimizer.update() — same trajectory as the in-kernel loop on the same draws"""
    torch.manual_seed(7)
    a = W.build_readme_ar(W.native_api(), T=20)
    inference.perform_inference(a, number_iterations=25, number_samples=300, optimizer=optimizer,
                                inference_method=_ReferenceStyleKL(), **kw)
    b = W.build_readme_ar(W.native_api(), T=20)
    inference.perform_inference(b, number_iterations=25, number_samples=300, optimizer=optimizer,
                                inference_method=inference.ReverseKL(), **kw)
    la, lb = np.asarray(a.diagnostics["loss curve"]), np.asarray(b.diagnostics["loss curve"])
    assert la.shape == lb.shape == (25,)
    assert rel_err(la, lb) <= 5e-6
    pa = engine.compile_model(a, None, "pathwise").params.cpu().numpy()
    pb = engine.compile_model(b, None, "pathwise").params.cpu().numpy()
    assert np.abs(pa - pb).max() <= 2e-5


def test_hand_written_loop_with_optimizer_update():
    """loss = -ELBO; loss.backward(); optimizer.update() — `optimizers.py:69-73`; the ELBO improves"""
    model = W.build_beta_binomial(W.native_api(), n_obs=30)
    opt = ProbabilisticOptimizer(model.posterior_model, "Adam", lr=0.05)
    values = []
    for _ in range(60):
        loss = -model.estimate_log_model_evidence(number_samples=512, for_gradient=True)
        opt.zero_grad()
        loss.backward()
        opt.update()
        values.append(float(loss))
    assert np.isfinite(values).all()
    assert np.mean(values[-10:]) < np.mean(values[:10])


def test_device_binding_without_set_device():
    """config.set_device('cuda:1') alone (no torch.cuda.set_device): tables, buffers, stream and launches on cuda:1"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from brancher_amd import config
    config.set_device("cuda:1")
    try:
        c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
        losses, finite = c.train(20, 300, "SGD", seed=1, lr=1e-3)
        assert losses.device.index == 1 and bool(finite.all())
    finally:
        config.set_device("cuda:0")


def test_logit_normal_variable_matches_the_reference_fixture():
    """README.md:30,56 writes the AR coefficient as ``LogitNormalVariable`` and multiplies with it; this package's
    LogitNormalVariable (a Normal on the logit scale whose arithmetic goes through sigmoid) gives the model of the
    reference fixture, which the reference itself can only express as NormalVariable + BF.sigmoid"""
    g = Golden("readme_ar_T20_N300")
    model = W.build_readme_ar(W.native_api(), logit_normal=True, **g.meta["kwargs"])
    for estimator in ("pathwise", "blackbox"):
        c = engine.compile_model(model, None, estimator)
        res = c.evaluate(g.N, noise=g.noise, minibatch=g.minibatch)
        ref = float(g.data["loss_" + estimator])
        assert abs(float(res["loss"].item()) - ref) <= TOL * abs(ref)
        if estimator == "pathwise":
            grad_check(c.named_grads(), g.group("grad_%s/" % estimator), TOL)
        else:       # (the yardstick of tests/test_gpu_parity.py: the double-precision oracle on the fixture's draws)
            yardstick_grad_check(c.named_grads(), exact_oracle(g, g.N, estimator, g.noise)["grads"], g.group("grad_%s/" % estimator))
