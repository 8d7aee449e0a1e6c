"""The exact-data iteration of the dense-link path as two fused launches (csrc/dense_xfused.inc: dense_xfwd draws the
weights in LDS beside the logits product and takes the cross-entropy in its epilogue, dense_xbwd reduces the gradient
product against the redrawn normals) against
  (1) the round-3 sequence of six launches on the same Philox stream and minibatch (BSVI_DENSE_FUSED=0), and
  (2) the f32-input MFMA kernels (BSVI_DENSE_XGEMM=0),
over shapes that exercise every edge of the tiling: classes that do not divide the 64 weight rows of a workgroup, a
feature count that is not a multiple of the 128-feature chunk or of the 112-row tile of x^T, minibatches below and above
one 512-row tile, sample counts that leave ragged sample groups, the Bernoulli likelihood (one output), both estimators,
emitted and supplied noise.  Reference behaviour: examples/MNIST_logistic_regression.py:15-54 through
brancher/variables.py:843-870 (pinned against the reference itself by the `logreg_pixels_*` fixtures in test_gpu_parity)."""
import numpy as np
import pytest
import torch

from brancher_amd import engine, workloads as W

pytestmark = pytest.mark.gpu

SHAPES = [
    # (dataset, batch, features, classes, samples)
    (96, 40, 784, 10, 24),
    (300, 130, 100, 3, 70),       # two k chunks short of full, one 112-row tile of x^T, ragged sample block
    (700, 520, 64, 16, 33),       # two 512-row tiles of the minibatch, 16 classes (four samples per workgroup)
    (200, 64, 236, 7, 130),       # three tiles of x^T with a ragged last one, 130 samples
    (64, 32, 32, 1, 50),          # smallest feature count of the path
]


def build(api, shape, **kw):
    ds, b, p, c, _ = shape
    return W.build_logistic_regression(api, dataset_size=ds, batch_size=b, n_features=p, n_classes=c, pixels="uint8", q_scale=0.02, **kw)


def outputs(c, n, **kw):
    res = c.evaluate(n, seed=13, offset=2, want_fvalues=True, want_indices=True, **kw)
    torch.cuda.synchronize()
    return dict(loss=float(res["loss"].item()), f=res["f"].cpu().numpy().astype(np.float64),
                grads=res["grads"].cpu().numpy().astype(np.float64), idx=res["indices"].cpu().numpy(),
                noise=res["noise"].cpu().numpy() if "noise" in res and res["noise"] is not None else None)


@pytest.mark.parametrize("estimator", ["pathwise", "blackbox"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "DS%d_B%d_P%d_C%d_N%d" % s)
def test_fused_launches_equal_the_six_launch_sequence_and_the_f32_kernels(shape, estimator, monkeypatch):
    api = W.native_api()
    n = shape[4]
    fused = engine.compile_model(build(api, shape), None, estimator)
    assert fused.data_path() == "bf16x3"
    monkeypatch.setenv("BSVI_DENSE_FUSED", "0")
    six = engine.compile_model(build(api, shape), None, estimator)
    monkeypatch.setenv("BSVI_DENSE_XGEMM", "0")
    plain = engine.compile_model(build(api, shape), None, estimator)
    assert plain.data_path() == "f32"
    a, b, c = outputs(fused, n, want_noise=True), outputs(six, n, want_noise=True), outputs(plain, n)
    assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["idx"], c["idx"])
    assert np.array_equal(a["noise"], b["noise"])                  # the same Philox counters
    fscale, gscale = np.abs(c["f"]).max(), max(np.abs(c["grads"]).max(), 1e-3)     # (one class: the likelihood's gradient is zero)
    bb = estimator == "blackbox"
    for other in (b, c):
        assert np.abs(a["f"] - other["f"]).max() <= 2e-5 * fscale
        assert abs(a["loss"] - other["loss"]) <= (2e-4 if bb else 2e-5) * abs(other["loss"])
        assert np.abs(a["grads"] - other["grads"]).max() <= (5e-4 if bb else 5e-5) * gscale + 1e-6
    # bit-reproducible call to call, with and without the diagnostic outputs
    again = outputs(fused, n, want_noise=True)
    assert again["loss"] == a["loss"] and np.array_equal(again["grads"], a["grads"]) and np.array_equal(again["f"], a["f"])
    # the emitted noise, fed back, reproduces the call (supplied-noise path of both launches)
    ds, bsz, p, ncls, _ = shape
    named = {"weights": a["noise"].T.reshape(n, 1, ncls, p)}
    replay = fused.evaluate(n, noise=named, minibatch={"indices": a["idx"].tolist()}, want_fvalues=True)
    torch.cuda.synchronize()
    assert np.abs(replay["f"].cpu().numpy() - a["f"]).max() <= 2e-6 * fscale
    assert np.abs(replay["grads"].cpu().numpy() - a["grads"]).max() <= 2e-6 * gscale


def test_fused_training_walks_the_trajectory_of_the_six_launch_sequence(monkeypatch):
    api = W.native_api()
    shape = (256, 64, 64, 10, 128)
    fused = engine.compile_model(build(api, shape), None, "pathwise")
    monkeypatch.setenv("BSVI_DENSE_FUSED", "0")
    six = engine.compile_model(build(api, shape), None, "pathwise")
    la, fa = fused.train(40, 128, "Adam", lr=5e-3, seed=3)
    lb, fb = six.train(40, 128, "Adam", lr=5e-3, seed=3)
    assert bool(fa.all()) and bool(fb.all())
    np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=2e-5)
    np.testing.assert_allclose(fused.params.cpu().numpy(), six.params.cpu().numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[3]], ids=lambda s: "DS%d_B%d_P%d_C%d_N%d" % s)
def test_sample_sums_inside_the_parameter_kernel_equal_the_epilogue_launch(shape, monkeypatch):
    """dense_param_fold_kernel (every workgroup adds the value of the estimator itself) against dense_epilogue + dense_param_kernel
    (BSVI_DENSE_FOLD=0): the same gradients bit for bit — they never pass through the folded sums — and the same value up to the
    association of a 256- against a 1024-thread sum; the optimizer steps walk the same trajectory."""
    api = W.native_api()
    n = shape[4]
    folded = engine.compile_model(build(api, shape), None, "pathwise")
    monkeypatch.setenv("BSVI_DENSE_FOLD", "0")
    two = engine.compile_model(build(api, shape), None, "pathwise")
    ra, rb = folded.evaluate(n, seed=5, offset=1), two.evaluate(n, seed=5, offset=1)
    torch.cuda.synchronize()
    assert torch.equal(ra["grads"], rb["grads"])
    assert abs(float(ra["loss"]) - float(rb["loss"])) <= 1e-6 * abs(float(rb["loss"]))
    la, fa = folded.train(30, n, "Adam", lr=5e-3, seed=3)
    lb, fb = two.train(30, n, "Adam", lr=5e-3, seed=3)
    assert bool(fa.all()) and bool(fb.all())
    np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=1e-6)
    assert torch.equal(folded.params, two.params)
