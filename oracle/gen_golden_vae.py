"""
ORACLE TOOLING — generates tests/golden/vae_*.npz by running the REAL reference (container only).

Same protocol as oracle/gen_golden.py, for the amortised workload of BASELINE config 5
(`examples/VAE_playground.py:18-88`, built by brancher_amd/workloads.py:build_vae through the reference's own
constructors): the reference's `ReverseKL.compute_loss -> estimate_log_model_evidence -> GradientEstimator ->
loss.backward()` runs on PyTorch-CPU with both estimators, and the only interception is recording — the rows
`np.random.choice` hands to `EmpiricalDistribution._get_sample` (`distributions.py:436-441`) and the raw
`_standard_normal` draw behind the latent's `rsample` — so that the HIP kernels and oracle/vae_oracle.py can be
fed the same minibatches and noise.  Parameters are the tensors of the two torch modules, keyed
"enc/<name>" / "dec/<name>".

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_vae.py [case ...]
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from gen_golden import DrawRecorder, reference_api, OUT  # noqa: E402

CASES = {
    # name: (builder kwargs, N, seed, trajectory)
    "vae_P12_H8_H6_DS20_B5_N3": (dict(dataset_size=20, batch_size=5, n_features=12, hidden1=8, hidden2=6), 3, 21,
                                 dict(iters=5, n=4, optimizer="Adam", lr=1e-2)),
    # several GEMM tiles in every dimension: rows 9*16 = 144 > 128, widths 150 / 40 / 136 straddle the 128 tile
    "vae_P150_H136_H40_DS40_B16_N9": (dict(dataset_size=40, batch_size=16, n_features=150, hidden1=136, hidden2=40,
                                           latent_size=3), 9, 22, dict(iters=4, n=9, optimizer="Adam", lr=1e-3)),
    # the example's image width (784 = 6 x 128 + 16) with narrower hidden layers, to keep the fixture below 1 MB; the
    # full 784-256-512-(2,2) / 2-512-256-784 architecture is covered by the oracle-vs-HIP tests
    "vae_P784_H72_H40_DS24_B8_N2": (dict(dataset_size=24, batch_size=8, n_features=784, hidden1=72, hidden2=40),
                                    2, 23, None),
    # the pattern one notch wider: real-valued rows under a Normal likelihood with a constant scale (one number / one per
    # feature), and a prior whose loc and scale are learnable parameters of the joint model
    "vae_normal_P20_H12_H8_DS30_B6_N4": (dict(dataset_size=30, batch_size=6, n_features=20, hidden1=12, hidden2=8,
                                              likelihood="normal", likelihood_scale=0.7), 4, 24,
                                         dict(iters=5, n=4, optimizer="Adam", lr=1e-2)),
    "vae_learnable_prior_P16_H10_H8_DS24_B5_N5": (dict(dataset_size=24, batch_size=5, n_features=16, hidden1=10, hidden2=8,
                                                       latent_size=3, learnable_prior=True), 5, 25,
                                                  dict(iters=5, n=5, optimizer="Adam", lr=1e-2)),
    "vae_normal_learnable_prior_P140_H72_H40_DS40_B12_N6": (
        dict(dataset_size=40, batch_size=12, n_features=140, hidden1=72, hidden2=40, latent_size=4, likelihood="normal",
             likelihood_scale=[0.4 + 0.01 * j for j in range(140)], learnable_prior=True), 6, 26,
        dict(iters=4, n=6, optimizer="SGD", lr=1e-3)),
    # ... and a LEARNABLE likelihood scale (NormalVariable(decoder value, scale, learnable=True): a root of the joint model behind
    # softplus): one value for every feature, and one per feature
    "vae_normal_learnable_scale_P24_H12_H8_DS30_B6_N5": (
        dict(dataset_size=30, batch_size=6, n_features=24, hidden1=12, hidden2=8, likelihood="normal", likelihood_scale=0.8,
             learnable_likelihood_scale=True), 5, 27, dict(iters=5, n=5, optimizer="Adam", lr=1e-2)),
    "vae_normal_learnable_scales_P132_H40_H24_DS36_B10_N7": (
        dict(dataset_size=36, batch_size=10, n_features=132, hidden1=40, hidden2=24, latent_size=3, likelihood="normal",
             likelihood_scale=[0.5 + 0.004 * j for j in range(132)], learnable_likelihood_scale=True, learnable_prior=True), 7, 28,
        dict(iters=4, n=7, optimizer="Adam", lr=5e-3)),
    # ... or a second HEAD of the decoder (a heteroscedastic decoder: NormalVariable(decoder(z)["mean"], decoder(z)["sd"]))
    "vae_normal_decoder_scale_P28_H16_H10_DS30_B6_N5": (
        dict(dataset_size=30, batch_size=6, n_features=28, hidden1=16, hidden2=10, likelihood="normal", likelihood_scale="decoder"),
        5, 29, dict(iters=5, n=5, optimizer="Adam", lr=1e-2)),
    "vae_normal_decoder_scale_P136_H72_H40_DS40_B12_N6": (
        dict(dataset_size=40, batch_size=12, n_features=136, hidden1=72, hidden2=40, latent_size=3, likelihood="normal",
             likelihood_scale="decoder", learnable_prior=True), 6, 30, dict(iters=4, n=6, optimizer="Adam", lr=5e-3)),
}


class ChoiceRecorder:
    def __enter__(self):
        self.calls = []
        self._orig = np.random.choice

        def choice(*a, **k):
            out = self._orig(*a, **k)
            self.calls.append(np.array(out, dtype=np.int64))
            return out

        np.random.choice = choice
        return self

    def __exit__(self, *exc):
        np.random.choice = self._orig


def module_params(model):
    enc, dec = model.vae_modules
    out = {}
    for tag, m in (("enc", enc), ("dec", dec)):
        for k, p in m.named_parameters():
            out["%s/%s" % (tag, k)] = p
    # learnable roots of the joint model (a learnable prior): the tensor behind the RootVariable's ParameterModule
    for v in sorted(model.flatten(), key=lambda v: v.name):
        if type(v).__name__ == "RootVariable" and getattr(v, "learnable", False):
            (p,) = list(v.link.parameters())
            out["prior/" + v.name] = p
    return out


def evaluate(model, q, est, N, capture):
    import torch
    from brancher import inference
    with DrawRecorder() as rec, ChoiceRecorder() as ch:
        loss = inference.ReverseKL(gradient_estimator=est).compute_loss(model, q, None, N)
    z = capture["z"]
    by_name = {v.name: s for v, s in z.items()}
    B = by_name["x"].shape[1]
    idx = np.stack(ch.calls)
    assert idx.shape == (N, B), idx.shape
    Dz = by_name["z"].shape[-1]
    eps = [d for k, d in rec.draws if k == "normal" and d.numel() == N * B * Dz]
    assert len(eps) == 1
    eps = eps[0].reshape(N, B, Dz)
    enc_out = by_name["encoder_output"]
    assert torch.equal((enc_out["mean"] + eps * enc_out["sd"]).detach(), by_name["z"].detach())
    return loss, idx, eps.detach().numpy().copy(), z


def run_case(name, api):
    import torch
    from brancher import gradient_estimators as ge
    from brancher.optimizers import ProbabilisticOptimizer
    from brancher import inference
    import brancher_amd.workloads as W

    kwargs, N, seed, traj = CASES[name]
    model = W.build_vae(api, **kwargs)
    model.update_observed_submodel()
    q = model.posterior_model
    q._input_variables = sorted(q._input_variables, key=lambda v: v.name)     # (as in gen_golden.py: a reproducible walk)
    params = module_params(model)
    out = {"param/" + k: p.detach().numpy().copy() for k, p in params.items()}
    dataset = W.vae_data(kwargs["dataset_size"], kwargs["n_features"], kwargs.get("seed", 0),
                         real=kwargs.get("likelihood") == "normal")

    capture = {}
    orig = q._get_sample

    def cap(*a, **k):
        res = orig(*a, **k)
        capture["z"] = dict(res)
        return res

    q._get_sample = cap
    for est_name, est in (("pathwise", ge.PathwiseDerivativeEstimator), ("blackbox", ge.BlackBoxEstimator)):
        for p in params.values():
            p.grad = None
        torch.manual_seed(seed)
        np.random.seed(seed)
        loss, idx, eps, z = evaluate(model, q, est, N, capture)
        loss.backward()
        if est_name == "pathwise":
            out["minibatch/x"], out["noise/z"] = idx, eps
            by_name = {v.name: s for v, s in z.items()}
            x = by_name["x"].detach().numpy()
            assert np.array_equal(x[..., 0], dataset[idx][..., 0].astype(np.float32)), "recorded rows do not reproduce x"
            out["z/z"] = by_name["z"].detach().numpy().copy()
            lp = model.get_p_log_probabilities_from_q_samples(q_samples=z, empirical_samples={}, for_gradient=True, q_model=q)
            out["lp"] = lp.detach().numpy().copy()
            out["H"] = q._get_entropy(z, for_gradient=True).detach().numpy().copy()
            out["lq"] = q.calculate_log_probability(z).detach().numpy().copy()
        else:
            assert np.array_equal(out["minibatch/x"], idx) and np.array_equal(out["noise/z"], eps)
        out["loss_" + est_name] = np.float32(loss.detach().numpy())
        for k, p in params.items():
            out["grad_%s/%s" % (est_name, k)] = p.grad.detach().numpy().copy()

    # user-defined estimators (the GradientEstimator seam, gradient_estimators.py:17-26; workloads.custom_estimators) on the
    # same draws and minibatches
    for est_name, est in sorted(W.custom_estimators(ge).items()):
        for p in params.values():
            p.grad = None
        torch.manual_seed(seed)
        np.random.seed(seed)
        loss, idx, eps, _ = evaluate(model, q, est, N, capture)
        loss.backward()
        assert np.array_equal(out["minibatch/x"], idx) and np.array_equal(out["noise/z"], eps), "estimators drew different noise"
        out["loss_custom_" + est_name] = np.float32(loss.detach().numpy())
        for k, p in params.items():
            out["grad_custom_%s/%s" % (est_name, k)] = (np.zeros_like(out["param/" + k]) if p.grad is None
                                                        else p.grad.detach().numpy().copy())

    if traj is not None:
        # the loop of inference.py:77-108: posterior optimizer every iteration, model optimizer from the second on
        method = inference.ReverseKL(gradient_estimator=ge.PathwiseDerivativeEstimator)
        opt_kwargs = {k: v for k, v in traj.items() if k not in ("iters", "n", "optimizer")}
        opts = [ProbabilisticOptimizer(m, traj["optimizer"], **opt_kwargs) for m in (q, model)]
        assert all(o.optimizer for o in opts)
        torch.manual_seed(seed + 1000)
        np.random.seed(seed + 1000)
        losses, idx_seq, eps_seq = [], [], []
        for it in range(traj["iters"]):
            loss, idx, eps, _ = evaluate(model, q, ge.PathwiseDerivativeEstimator, traj["n"], capture)
            idx_seq.append(idx)
            eps_seq.append(eps)
            if torch.isfinite(loss.detach()).all().item():
                [o.zero_grad() for o in opts]
                loss.backward()
                opts[0].update()
                if it > 0:
                    [o.update() for o in opts[1:]]
            losses.append(float(loss.detach()))
        out["traj/losses"] = np.array(losses, dtype=np.float32)
        out["traj/minibatch/x"] = np.stack(idx_seq)
        out["traj/noise/z"] = np.stack(eps_seq)
        for k, p in params.items():
            out["traj/param_after/" + k] = p.detach().numpy().copy()

    meta = dict(case=name, builder="build_vae", kwargs=kwargs, N=N, seed=seed, trajectory=traj,
                torch=torch.__version__, numpy=np.__version__, posterior_order="sorted by name",
                reference="LucaAmbrogioni/Brancher @ /root/reference")
    out["meta"] = np.array(json.dumps(meta))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-34s loss_pathwise=%.6f loss_blackbox=%.6f  (%d arrays, %d KB)" % (
        name, out["loss_pathwise"], out["loss_blackbox"], len(out), os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024))


if __name__ == "__main__":
    api = reference_api()
    from brancher import standard_variables as sv
    api.BinomialVariable = sv.BinomialVariable
    for case in (sys.argv[1:] or list(CASES)):
        run_case(case, api)
