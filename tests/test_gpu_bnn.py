"""The Bayesian-neural-network family (bsvi_bnn_*, brancher_amd/bnn.py) — the reference's tests/test_MNIST_bayesian_neural_network.py:
latent weight matrices AND biases of every layer.  The three reference fixtures (tests/golden/bnn_*) go through the generic golden
tests of test_gpu_parity.py (loss, per-sample f, every gradient, both estimators, trajectories under Adam / SGD); here: the Philox
path at the example's full size against the oracle on the draws the kernels report, the exact-piece path against the f32-input
path, size-independent properties at BASELINE config 4's scale, and training through the public API."""
import numpy as np
import pytest
import torch

from conftest import Golden, exact_oracle, rel_err, yardstick_grad_check, yardstick_loss_check
from brancher_amd import engine, workloads as W
from oracle.svi_oracle import Oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5
EXAMPLE = dict(dataset_size=96, batch_size=30, n_features=784, n_hidden=20, n_classes=10, q_scale1=4e-4, q_loc_scale=1.0)


def compile_bnn(estimator="pathwise", **kw):
    c = engine.compile_model(W.build_bayesian_neural_network(W.native_api(), **kw), None, estimator)
    assert type(c).__name__ == "CompiledBnn"
    return c


@pytest.mark.parametrize("estimator", ["pathwise", "blackbox"])
@pytest.mark.parametrize("kw,n", [(EXAMPLE, 50),                                                        # the example: H = 20, B = 30, N = 50
                                  (dict(EXAMPLE, pixels="unit", q_scale1=0.05), 20),                   # data that is not bf16: the f32-input products
                                  (dict(dataset_size=40, batch_size=17, n_features=64, n_hidden=9, hidden2=6, n_classes=5, q_scale1=3e-3,
                                        q_loc_scale=1.0, activation="relu"), 70),                       # three layers, ragged sizes
                                  (dict(dataset_size=40, batch_size=33, n_features=36, n_hidden=5, n_classes=3, q_scale1=5e-3,
                                        q_loc_scale=1.0, activation="sigmoid"), 130),
                                  (dict(dataset_size=64, batch_size=24, n_features=48, n_hidden=7, n_classes=1, q_scale1=2e-3,
                                        q_loc_scale=1.0), 90)])                                          # one logit: Binomial(1, logits=...)
def test_philox_path_matches_the_oracle_on_the_reported_draws(kw, n, estimator):
    """noise and minibatch drawn on the device, reported by the kernels and replayed by the oracle in double precision; the bound is
    the suite's yardstick — as close to it as the reference arithmetic (the oracle in single precision) is (x4), or 1e-5 of the scale"""
    c = compile_bnn(estimator, **kw)
    assert c.data_path() == ("f32" if kw.get("pixels") == "unit" else "bf16x3")
    res = c.evaluate(n, seed=3, offset=5, want_noise=True, want_indices=True, want_fvalues=True)
    idx = res["indices"].cpu().numpy()
    assert len(set(idx.tolist())) == kw["batch_size"] and idx.min() >= 0 and idx.max() < kw["dataset_size"]
    named = c.named_noise(res["noise"].cpu().numpy(), n)
    mb = {c.program.indices_name: idx.tolist()}
    build = lambda: W.build_bayesian_neural_network(W.native_api(), **kw)
    exact = Oracle(build(), dtype=torch.float64).loss_and_grads(n, estimator, named, mb)
    ref32 = Oracle(build()).loss_and_grads(n, estimator, named, mb)
    # f = lp + H cancels: the value is measured against the summands as on the dense path, or against the single-precision oracle
    f64 = exact["f"].reshape(-1)
    summands = float(np.abs(f64).max()) + 0.5 * c.program.n_rows * 3.0
    assert np.abs(res["f"].cpu().numpy() - f64).max() <= max(4 * np.abs(ref32["f"].reshape(-1) - f64).max(), 1e-6 * summands)
    loss = float(res["loss"].item())
    assert abs(loss - exact["loss"]) <= max(4 * abs(ref32["loss"] - exact["loss"]), TOL * abs(exact["loss"]), 1e-6 * summands), \
        (loss, exact["loss"], ref32["loss"])
    yardstick_grad_check(c.named_grads(), exact["grads"], ref32["grads"])
    # the same draws fed back in: the noise-in path of every kernel (and bit-equal repeat calls)
    res2 = c.evaluate(n, noise=named, minibatch=mb, want_fvalues=True)
    assert abs(float(res2["loss"].item()) - loss) <= 2e-6 * max(1.0, abs(loss))
    g2 = c.named_grads()
    res3 = c.evaluate(n, noise=named, minibatch=mb, want_fvalues=True)
    assert torch.equal(res2["f"], res3["f"]) and all(np.array_equal(g2[k], v) for k, v in c.named_grads().items())


@pytest.mark.parametrize("estimator", ["pathwise", "blackbox"])
def test_exact_piece_products_equal_the_f32_input_products(estimator, monkeypatch):
    """pixel counts are exactly bf16: both products run as three bf16 MFMAs on the exact pieces of the other operand — f32
    semantics on the bf16 matrix cores; BSVI_DENSE_XGEMM=0 (at create time) sends the same model through the f32-input kernels"""
    kw = dict(dataset_size=200, batch_size=64, n_features=256, n_hidden=16, n_classes=10, q_scale1=1e-3, q_loc_scale=1.0)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("BSVI_DENSE_XGEMM", flag)
        c = compile_bnn(estimator, **kw)
        assert c.data_path() == ("bf16x3" if flag == "1" else "f32")
        res = c.evaluate(96, seed=9, offset=1, want_fvalues=True)
        out[flag] = (res["f"].cpu().numpy(), float(res["loss"].item()), c.named_grads())
    fa, la, ga = out["1"]
    fb, lb, gb = out["0"]
    assert np.abs(fa - fb).max() <= 2e-6 * (np.abs(fb).max() + 1500.0)
    assert abs(la - lb) <= (2e-5 if estimator == "pathwise" else 2e-4) * abs(lb)
    gscale = max(np.abs(v).max() for v in gb.values())
    for name in gb:
        assert np.abs(ga[name] - gb[name]).max() <= (5e-5 if estimator == "pathwise" else 5e-4) * gscale, name


def test_sums_are_linear_over_sample_shards_at_config4_scale():
    """BASELINE config 4's shape with the example's hidden layer (784 -> 20 -> 10, minibatch 512, 1024 samples): the output block of
    the whole shard is the sum of the blocks of two halves (the Philox counters are GLOBAL sample indices), call after call
    bit-identical, every value finite"""
    from brancher_amd.native import OUT_HEADER
    import ctypes as C
    from brancher_amd import native
    kw = dict(dataset_size=4096, batch_size=512, n_features=784, n_hidden=20, n_classes=10, q_scale1=4e-4, q_loc_scale=1.0)
    c = compile_bnn("pathwise", **kw)
    n = 1024
    blocks = []
    for base, n_local in ((0, n), (0, n // 2), (n // 2, n - n // 2)):
        args = c._args(n_local, n, base, seed=21, offset=4)
        native.check(c.lib.bsvi_bnn_fwd_bwd(c.handle, C.byref(args)))
        blocks.append(c.out.detach().cpu().numpy().copy())
        native.check(c.lib.bsvi_bnn_fwd_bwd(c.handle, C.byref(args)))
        assert np.array_equal(blocks[-1], c.out.detach().cpu().numpy())
    whole, a, b = blocks
    assert np.isfinite(whole).all() and whole[1] == 0.0
    scale = np.abs(whole[OUT_HEADER:]).max()
    assert scale > 0 and np.abs(a[OUT_HEADER:] + b[OUT_HEADER:] - whole[OUT_HEADER:]).max() <= 2e-5 * scale
    assert abs(a[0] + b[0] - whole[0]) <= 2e-6 * (abs(whole[0]) + 1024 * 16000.0)


def test_the_reference_example_trains_through_the_public_api():
    """`inference.perform_inference(model, number_iterations, number_samples=50, optimizer='Adam', lr=0.005)` as the reference's
    test file calls it (tests/test_MNIST_bayesian_neural_network.py:56-60), on synthetic pixel counts with labels that depend on
    the pixels: the loss curve is finite and falls"""
    from brancher_amd import inference
    api = W.native_api()
    model = W.build_bayesian_neural_network(api, dataset_size=256, batch_size=30, n_features=784, n_hidden=20, n_classes=10, q_scale1=4e-4,
                                            q_loc_scale=1.0)
    inference.perform_inference(model, number_iterations=80, number_samples=50, optimizer="Adam", lr=0.005)
    curve = np.asarray(model.diagnostics["loss curve"])
    assert len(curve) == 80 and np.all(np.isfinite(curve)) and curve[-15:].mean() < curve[:15].mean()
    assert type(engine.compile_model(model, None, "pathwise")).__name__ == "CompiledBnn"


def test_two_rank_step_sequence_equals_the_fused_step():
    """bsvi_bnn_fwd_bwd + bsvi_finalize_step (what a rank of the sharded path runs around its all-reduce) against bsvi_bnn_step"""
    kw = dict(dataset_size=64, batch_size=20, n_features=64, n_hidden=8, n_classes=4, q_scale1=3e-3, q_loc_scale=1.0)
    a, b = compile_bnn("pathwise", **kw), compile_bnn("pathwise", **kw)
    la, _ = a.train(6, 40, "Adam", seed=2, lr=1e-2)
    lb, _ = b.train(6, 40, "Adam", seed=2, lr=1e-2, _force_sharded_path=True)
    # (the fused launch divides by N where bsvi_finalize_step multiplies by 1 / N: equal to rounding, as on the dense path)
    assert a.last_mode == "stepwise" and b.last_mode == "stepwise"
    np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(b.params.cpu().numpy(), a.params.cpu().numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("estimator", ["pathwise", "blackbox"])
@pytest.mark.parametrize("kw,n", [(EXAMPLE, 50),
                                  (dict(dataset_size=40, batch_size=17, n_features=64, n_hidden=9, hidden2=6, n_classes=5, q_scale1=3e-3,
                                        q_loc_scale=1.0, activation="relu"), 70),
                                  (dict(dataset_size=64, batch_size=24, n_features=48, n_hidden=7, n_classes=1, q_scale1=2e-3,
                                        q_loc_scale=1.0), 90)])
def test_generated_middle_equals_the_separate_launches(kw, n, estimator, monkeypatch):
    """Round 5: on exact data the upper layers, the likelihood, the reverse sweep, the pieces of d f / d a1 and the gradients of the small
    tensors are ONE kernel generated for the network (bnn_mid_gen: a workgroup per sample, lanes along the minibatch rows), and the
    per-(sample, row) kernel of the other data path is generated too (bnn_upper_gen).  Both against the generic kernels they replace
    (BSVI_BNN_MID=0 / BSVI_BNN_JIT=0: bnn_upper + bnn_small + bnn_lik) on the same draws: the same arithmetic per (sample, row), sums
    over the minibatch in another association — agreement to rounding; each form bit-reproducible."""
    def outputs():
        c = compile_bnn(estimator, **kw)
        res = c.evaluate(n, seed=9, offset=2, want_fvalues=True)
        torch.cuda.synchronize()
        again = c.evaluate(n, seed=9, offset=2, want_fvalues=True)
        torch.cuda.synchronize()
        assert torch.equal(res["f"], again["f"]) and torch.equal(res["grads"], again["grads"])
        return float(res["loss"]), res["f"].cpu().numpy().astype(np.float64), res["grads"].cpu().numpy().astype(np.float64)

    gen = outputs()                                    # bnn_mid_gen
    monkeypatch.setenv("BSVI_BNN_MID", "0")
    upper_gen = outputs()                              # bnn_upper_gen + bnn_small + bnn_lik
    monkeypatch.setenv("BSVI_BNN_JIT", "0")
    generic = outputs()                                # bnn_upper + bnn_small + bnn_lik
    fscale, gscale = np.abs(generic[1]).max(), max(np.abs(generic[2]).max(), 1e-6)
    bb = estimator == "blackbox"
    for other in (gen, upper_gen):
        assert abs(other[0] - generic[0]) <= (2e-4 if bb else 2e-6) * abs(generic[0])
        assert np.abs(other[1] - generic[1]).max() <= 2e-6 * fscale
        assert np.abs(other[2] - generic[2]).max() <= (2e-4 if bb else 5e-6) * gscale
