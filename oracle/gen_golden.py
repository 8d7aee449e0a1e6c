"""
ORACLE TOOLING — generates tests/golden/*.npz by running the REAL reference.

Runs only in the build container: it imports LucaAmbrogioni/Brancher from /root/reference
(read-only, never copied) on PyTorch-CPU, builds the workloads of brancher_amd/workloads.py
through the reference's own constructors, and drives the reference's own
`ReverseKL.compute_loss -> estimate_log_model_evidence -> GradientEstimator.__call__ ->
loss.backward()` (`brancher/inference.py:140-144`, `variables.py:843-870`,
`gradient_estimators.py:29-44`) and optimizer loop (`inference.py:95-108`, re-driven by the
12-line harness of SURVEY Appendix B because `perform_inference` itself raises at
`inference.py:109` under numpy >= 1.24).

The only interception is *recording*: the raw random draws torch makes (`_standard_normal`,
`Tensor.cauchy_`, `Tensor.uniform_`) and the dictionary returned by
`posterior_model._get_sample` are captured so that the same noise can be fed to the HIP
kernel and to oracle/svi_oracle.py.  Nothing in the reference's arithmetic is altered.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [case ...]
"""
import json
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.environ.get("BSVI_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # (tests regenerate into a temp dir)

CASES = {
    # name: (builder, builder kwargs, N, seed, trajectory spec)
    "gp_regression_n8_N50": ("build_gp_regression", dict(n=8), 50, 6, dict(iters=5, n=16, optimizer="Adam", lr=2e-2)),
    "readme_ar_T20_N300": ("build_readme_ar", dict(T=20), 300, 0, dict(iters=6, n=64, optimizer="SGD", lr=1e-3)),
    "readme_ar_T5_N7": ("build_readme_ar", dict(T=5), 7, 1, dict(iters=5, n=7, optimizer="Adam", lr=5e-2)),
    "readme_ar_T200_N32": ("build_readme_ar", dict(T=200), 32, 2, None),
    # scales of q that are sampled latents: Taylor1 evaluates those entropies on the parents' means
    "scale_from_latent_N60": ("build_scale_from_latent", dict(n_obs=6), 60, 31, dict(iters=4, n=30, optimizer="Adam", lr=1e-2)),
    "beta_ar_T20_N100": ("build_beta_ar", dict(T=20), 100, 3, dict(iters=4, n=32, optimizer="Adam", lr=1e-2)),
    "beta_binomial_N512": ("build_beta_binomial", dict(n_obs=30), 512, 4, dict(iters=6, n=128, optimizer="SGD", lr=0.1)),
    "lognormal_normal_N100": ("build_lognormal_normal", dict(n_obs=20), 100, 5,
                              dict(iters=5, n=50, optimizer="SGD", lr=1e-4)),
    "observed_ar_T50_N40": ("build_observed_ar", dict(T=50, q_concentration=2.0), 40, 10, dict(iters=4, n=30, optimizer="Adam", lr=0.05)),
    "multivariate_regression_n100_N50": ("build_multivariate_regression", dict(n=100), 50, 9,
                                         dict(iters=5, n=40, optimizer="Adam", lr=1e-3)),
    "dynamic_causal_model_T30_N20": ("build_dynamic_causal_model", dict(steps=30), 20, 16,
                                     dict(iters=4, n=5, optimizer="Adam", lr=0.01)),
    # an nn.Module as a link on the scalar path (functions.py:15-41): its tensors are learnable parameters of the joint model
    "module_link_mlp_N40": ("build_module_link_regression", dict(n_obs=6, hidden=4), 40, 41, dict(iters=5, n=24, optimizer="Adam", lr=1e-2)),
    # ... with several output units (round 6: a vector-valued module output, one model term per unit)
    "module_link_mlp_in2_out3_N30": ("build_module_link_regression", dict(n_obs=5, hidden=4, n_in=2, n_out=3), 30, 43,
                                     dict(iters=4, n=16, optimizer="Adam", lr=1e-2)),
    "map_estimate_N3": ("build_map_estimate", dict(n_obs=12), 3, 15, dict(iters=6, n=2, optimizer="SGD", lr=0.01)),
    "vector_latent_d4_N70": ("build_vector_latent", dict(n_obs=9, dim=4), 70, 14, dict(iters=4, n=33, optimizer="SGD", lr=1e-3)),
    "linear_predictor_d4_N40": ("build_linear_predictor", dict(n_obs=5, dim=4), 40, 17, dict(iters=4, n=24, optimizer="Adam", lr=1e-2)),
    "flat_vector_sum_d5_N48": ("build_flat_vector_sum", dict(n_obs=6, dim=5), 48, 29, dict(iters=4, n=24, optimizer="Adam", lr=1e-2)),
    "softmax_classifier_C3_N60": ("build_softmax_classifier", dict(n_obs=6, n_classes=3), 60, 19, dict(iters=4, n=32, optimizer="Adam", lr=1e-2)),
    "gp_hyperparameters_n5_N80": ("build_gp_hyperparameters", dict(n=5), 80, 23, dict(iters=5, n=32, optimizer="Adam", lr=1e-2)),
    # the same model with 32 / 100 inputs: the covariance no longer fits the per-sample program (batched kernel, bsvi_mvn_*)
    "gp_hyperparameters_n32_N40": ("build_gp_hyperparameters", dict(n=32, jitter=5e-2), 40, 31, dict(iters=4, n=16, optimizer="Adam", lr=1e-2)),
    "gp_hyperparameters_n100_N24": ("build_gp_hyperparameters", dict(n=100, jitter=5e-2), 24, 37, dict(iters=3, n=12, optimizer="Adam", lr=1e-2)),
    # a posterior whose mean of f follows a SAMPLED parent: under Taylor1 the value of the multivariate-normal term is an expression
    # of that draw (a per-sample mean, rows of the draw behind the posterior's own)
    "gp_structured_mean_n12_N40": ("build_gp_hyperparameters", dict(n=12, jitter=5e-2, structured_mean=True), 40, 67, dict(iters=4, n=16, optimizer="Adam", lr=1e-2)),
    "gp_structured_mean_n48_N24": ("build_gp_hyperparameters", dict(n=48, jitter=5e-2, structured_mean=True), 24, 71, dict(iters=3, n=12, optimizer="Adam", lr=1e-2)),
    # the function values integrated out (an OBSERVED MultivariateNormal): 40 inputs, and 200 / 260 — beyond what LDS holds, the
    # batched kernel keeps the matrix of a sample in device memory (MVN_SPILL)
    "gp_marginal_n40_N32": ("build_gp_marginal_likelihood", dict(n=40), 32, 53, dict(iters=4, n=16, optimizer="Adam", lr=1e-2)),
    "gp_marginal_n200_N16": ("build_gp_marginal_likelihood", dict(n=200), 16, 59, dict(iters=3, n=8, optimizer="Adam", lr=1e-2)),
    "gp_marginal_n260_N12": ("build_gp_marginal_likelihood", dict(n=260), 12, 61, dict(iters=3, n=6, optimizer="Adam", lr=1e-2)),
    # the other two parameterisations of the MultivariateNormal node, with a matrix that depends on a sampled scale
    # (bsvi_mvn_form: the batched kernel skips the factorisation for a scale_tril and factorises the precision in its place)
    "mvn_scale_tril_n24_N40": ("build_mvn_forms", dict(n=24, form="scale_tril"), 40, 41, dict(iters=4, n=16, optimizer="Adam", lr=1e-2)),
    "mvn_precision_n24_N40": ("build_mvn_forms", dict(n=24, form="precision_matrix"), 40, 43, dict(iters=4, n=16, optimizer="Adam", lr=1e-2)),
    "mvn_precision_n6_N60": ("build_mvn_forms", dict(n=6, form="precision_matrix"), 60, 47, None),
    "learnable_model_N60": ("build_learnable_model", dict(n_obs=15), 60, 13, dict(iters=6, n=40, optimizer="Adam", lr=0.02)),
    "discrete_latent_N200": ("build_discrete_latent", dict(n_obs=8), 200, 12, None),
    "heavy_tails_N64": ("build_heavy_tails", dict(n_obs=12), 64, 6, dict(iters=4, n=32, optimizer="Adam", lr=1e-2)),
    # dense matmul link + Categorical likelihood + random minibatch (BASELINE config 4, reduced sizes)
    "logreg_C3_P6_DS20_B12_N5": ("build_logistic_regression",
                                 dict(dataset_size=20, batch_size=12, n_features=6, n_classes=3), 5, 7,
                                 dict(iters=5, n=6, optimizer="Adam", lr=5e-3)),
    # examples/minibatch_logistic_regression.py: Binomial(1, logits) likelihood, labels through a shared minibatch
    "logreg_binary_P2_DS50_B30_N6": ("build_binary_logistic_regression", dict(dataset_size=50, batch_size=30, n_features=2),
                                     6, 11, dict(iters=5, n=8, optimizer="Adam", lr=0.05)),
    "logreg_map_C3_P4_DS30_N2": ("build_map_logistic_regression", dict(dataset_size=30, n_features=4, n_classes=3), 2, 17,
                                 dict(iters=5, n=1, optimizer="SGD", lr=0.0025)),
    # the same with pixel COUNTS 0..255 as the example feeds them (exactly bf16 numbers: the dense path's bf16 matrix-core products)
    "logreg_pixels_C10_P784_DS24_B16_N4": ("build_logistic_regression",
                                           dict(dataset_size=24, batch_size=16, n_features=784, n_classes=10, pixels="uint8", q_scale=0.01),
                                           4, 9, None),
    "logreg_pixels_C3_P64_DS40_B24_N6": ("build_logistic_regression",
                                         dict(dataset_size=40, batch_size=24, n_features=64, n_classes=3, pixels="uint8", q_scale=0.02),
                                         6, 10, dict(iters=4, n=5, optimizer="SGD", lr=1e-7)),     # (SGD: Adam's first steps are the SIGN of gradients that cancel to rounding noise here)
    "logreg_C10_P784_DS24_B16_N4": ("build_logistic_regression",
                                    dict(dataset_size=24, batch_size=16, n_features=784, n_classes=10), 4, 8, None),
    # (round 6) the minibatch data path OUTSIDE the matmul patterns of the dense / BNN / amortised families: observations that are rows of
    # a dataset on the scalar engine — a Normal mean with a latent noise scale, and minibatched Bayesian linear regressions (Normal
    # likelihood over BF.matmul(weights, x)) with a Normal / Laplace prior and with a latent prior scale shared by the weights
    "minibatch_normal_mean_DS40_B8_N30": ("build_minibatch_normal_mean", dict(dataset_size=40, batch_size=8), 30, 81,
                                          dict(iters=5, n=12, optimizer="Adam", lr=0.02)),
    "minibatch_linreg_P3_DS40_B8_N30": ("build_minibatch_linear_regression", dict(dataset_size=40, batch_size=8, n_features=3), 30, 83,
                                        dict(iters=5, n=12, optimizer="Adam", lr=0.02)),
    "minibatch_linreg_laplace_P4_O2_DS30_B6_N24": ("build_minibatch_linear_regression",
                                                   dict(dataset_size=30, batch_size=6, n_features=4, n_outputs=2, prior="laplace"), 24, 85,
                                                   dict(iters=4, n=10, optimizer="SGD", lr=1e-3)),
    "minibatch_linreg_tau_P3_DS40_B8_N30": ("build_minibatch_linear_regression",
                                            dict(dataset_size=40, batch_size=8, n_features=3, latent_scale=True), 30, 87,
                                            dict(iters=5, n=12, optimizer="Adam", lr=0.02)),
    # (round 6) examples/PopulationReceptiveFields.py at the example's size: a double BF.sum over a 40 x 40 mesh per datapoint and sample, and
    # an "observed" stimulus node the reference draws once per evaluation (recorded under drawn/)
    "prf_F40_D15_N20": ("build_population_receptive_fields", dict(field=40, n_data=15), 20, 91,
                        dict(iters=3, n=10, optimizer="Adam", lr=0.01)),
    # the reference's Bayesian neural network (tests/test_MNIST_bayesian_neural_network.py:20-60): latent weight matrices AND biases of
    # both layers, tanh hidden units, observed Categorical over a random minibatch — reduced sizes, one at the example's full width
    "bnn_P48_H6_C4_DS30_B12_N5": ("build_bayesian_neural_network",
                                  dict(dataset_size=30, batch_size=12, n_features=48, n_hidden=6, n_classes=4, q_scale1=2e-3, q_loc_scale=1.0),
                                  5, 21, dict(iters=4, n=4, optimizer="Adam", lr=5e-3)),
    "bnn_P784_H20_C10_DS40_B30_N3": ("build_bayesian_neural_network",
                                     dict(dataset_size=40, batch_size=30, n_features=784, n_hidden=20, n_classes=10, q_scale1=4e-4, q_loc_scale=1.0),
                                     3, 22, None),
    "bnn_relu_P32_H8_H5_C3_DS24_B10_N6": ("build_bayesian_neural_network",
                                          dict(dataset_size=24, batch_size=10, n_features=32, n_hidden=8, hidden2=5, n_classes=3, q_scale1=3e-3,
                                               q_loc_scale=1.0, activation="relu"), 6, 23, dict(iters=3, n=4, optimizer="SGD", lr=1e-3)),
}


# workloads whose posterior is made of Normal variables: the Taylor1 estimator is recorded for them too
TAYLOR1_BUILDERS = ("build_readme_ar", "build_multivariate_regression", "build_learnable_model", "build_vector_latent", "build_scale_from_latent",
                    "build_beta_binomial", "build_observed_ar", "build_lognormal_normal",
                    # (round 4: models with a MultivariateNormal term whose matrix depends on a latent — the taylor1 program reads it
                    #  at the posterior's means)
                    "build_gp_hyperparameters", "build_mvn_forms", "build_gp_marginal_likelihood",
                    # (round 6: the Bayesian neural network — every latent a mean-field Normal, its mean its loc)
                    "build_bayesian_neural_network")


# ... and two user-defined estimators (workloads.custom_estimators) for these
CUSTOM_ESTIMATOR_BUILDERS = ("build_readme_ar", "build_vector_latent", "build_heavy_tails",
                             # ... and the dense-link path (BASELINE config 4 at reduced sizes)
                             "build_logistic_regression", "build_binary_logistic_regression",
                             # ... and the Bayesian-neural-network path (round 6)
                             "build_bayesian_neural_network",
                             # ... and minibatch observations on the scalar path (round 6)
                             "build_minibatch_normal_mean", "build_minibatch_linear_regression")


def reference_api():
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    warnings.filterwarnings("ignore")
    from brancher import standard_variables as sv, variables as v, functions as BF
    return types.SimpleNamespace(
        NormalVariable=sv.NormalVariable, LogNormalVariable=sv.LogNormalVariable, BetaVariable=sv.BetaVariable,
        BinomialVariable=sv.BinomialVariable, BernulliVariable=sv.BernulliVariable,
        CauchyVariable=sv.CauchyVariable, LaplaceVariable=sv.LaplaceVariable,
        DeterministicVariable=sv.DeterministicVariable, RootVariable=v.RootVariable,
        CategoricalVariable=sv.CategoricalVariable, EmpiricalVariable=sv.EmpiricalVariable,
        RandomIndices=sv.RandomIndices, MultivariateNormalVariable=sv.MultivariateNormalVariable,
        ProbabilisticModel=v.ProbabilisticModel, BF=BF, name="reference")


class DrawRecorder:
    """Records every raw random draw torch makes while active."""

    def __enter__(self):
        import torch
        import torch.distributions.normal as tn
        self.torch, self.tn = torch, tn
        self.draws = []
        self._sn = tn._standard_normal
        self._cauchy = torch.Tensor.cauchy_
        self._uniform = torch.Tensor.uniform_
        rec = self.draws

        def standard_normal(shape, dtype, device):
            out = self._sn(shape, dtype=dtype, device=device)
            rec.append(("normal", out.clone()))
            return out

        def cauchy_(t, *a, **k):
            out = self._cauchy(t, *a, **k)
            rec.append(("cauchy", out.clone()))
            return out

        def uniform_(t, *a, **k):
            out = self._uniform(t, *a, **k)
            rec.append(("uniform", out.clone()))
            return out

        tn._standard_normal = standard_normal
        torch.Tensor.cauchy_ = cauchy_
        torch.Tensor.uniform_ = uniform_
        return self

    def __exit__(self, *exc):
        self.tn._standard_normal = self._sn
        self.torch.Tensor.cauchy_ = self._cauchy
        self.torch.Tensor.uniform_ = self._uniform


def match_noise(q, z, draws):
    """For every random posterior variable find the raw draw that reproduces its sample
    bit-for-bit through torch's own rsample formula; Beta/discrete draws are their own noise."""
    import torch
    from brancher import distributions as rd
    noise = {}
    for var, sample in z.items():
        dist = getattr(var, "distribution", None)
        if isinstance(dist, (rd.DeterministicDistribution,)) or dist is None:
            continue
        if type(var).__name__ == "RootVariable":
            continue
        params = var._get_parameters_from_input_values(z)
        if isinstance(dist, (rd.NormalDistribution, rd.LogNormalDistribution, rd.CauchyDistribution,
                             rd.LaplaceDistribution)):
            loc, scale = params["loc"], params["scale"]
            # (the reference pads parameters with TRAILING singleton axes up to the sample's rank, utilities.py:143-159)
            loc = loc.reshape(tuple(loc.shape) + (1,) * (sample.dim() - loc.dim()))
            scale = scale.reshape(tuple(scale.shape) + (1,) * (sample.dim() - scale.dim()))
            kind = {"NormalDistribution": "normal", "LogNormalDistribution": "normal",
                    "CauchyDistribution": "cauchy", "LaplaceDistribution": "uniform"}[type(dist).__name__]
            found = None
            for k, d in draws:
                if k != kind or d.numel() != sample.numel():
                    continue
                d4 = d.reshape(sample.shape)
                if kind == "uniform":
                    rec = loc - scale * d4.sign() * torch.log1p(-d4.abs())
                else:
                    rec = loc + d4 * scale
                    if isinstance(dist, rd.LogNormalDistribution):
                        rec = rec.exp()
                if torch.equal(rec.detach(), sample.detach()):
                    found = d4
                    break
            if found is None:
                raise RuntimeError("could not match the noise of %s" % var.name)
            noise[var.name] = found.detach().numpy().copy()
        else:
            noise[var.name] = sample.detach().numpy().copy()
    return noise


class _ModuleTensor:
    """A tensor of an nn.Module used as a link (functions.py:15-20), recorded like a learnable root: `.link.parameter` is the
    nn.Parameter the reference's optimizer steps (optimizers.py:36-49), `.value` its current value."""

    def __init__(self, tensor):
        self.link = types.SimpleNamespace(parameter=tensor)

    @property
    def value(self):
        return self.link.parameter


def named_parameters(model, q):
    out = {}
    for m in (q, model):
        for v in m.flatten():
            if type(v).__name__ == "RootVariable" and v.learnable:
                out.setdefault(v.name, v)
    # (builders that use module links name them: {"net": module} -> "net.0.weight", ... as brancher_amd names the tensors)
    for fname, module in getattr(model, "_golden_modules", {}).items():
        for pname, tensor in module.named_parameters():
            out.setdefault("%s.%s" % (fname, pname), _ModuleTensor(tensor))
    return out


def run_case(name, api):
    import torch
    from brancher import inference, gradient_estimators as ge
    from brancher.optimizers import ProbabilisticOptimizer
    import brancher_amd.workloads as W

    builder, kwargs, N, seed, traj = CASES[name]
    model = getattr(W, builder)(api, **kwargs)
    model.update_observed_submodel()
    q = model.posterior_model
    # The reference walks `posterior_model.variables`, a SET of objects hashed by address (variables.py:66,636,894), so
    # which torch draw goes to which variable changes from run to run.  The fixtures record the noise per variable NAME,
    # which makes each of them self-consistent either way; walking the variables in name order makes regenerating one
    # reproducible as well (recorded in meta["posterior_order"]).
    q._input_variables = sorted(q._input_variables, key=lambda v: v.name)
    roots = named_parameters(model, q)
    out = {}
    for pname, root in roots.items():
        out["param/" + pname] = root.value.detach().numpy().copy()

    captured = {}
    orig = q._get_sample

    def capture(*a, **k):
        res = orig(*a, **k)
        captured["z"] = dict(res)
        return res

    q._get_sample = capture

    # minibatch indices drawn by RandomIndices variables (numpy RNG) are recorded the same way
    obs_model = model.observed_submodel
    orig_obs = obs_model._get_sample

    def capture_obs(*a, **k):
        res = orig_obs(*a, **k)
        captured["minibatch"] = {var.name: np.array([int(i) for i in val], dtype=np.int64)
                                 for var, val in res.items() if type(var).__name__ == "RandomIndices"}
        captured["emp"] = dict(res)
        # variables observed by flag only (no value): drawn from their own distribution once per evaluation (variables.py:553-565)
        captured["drawn"] = {var.name: val.detach().numpy().copy() for var, val in res.items()
                             if getattr(var, "is_observed", False) and hasattr(var, "has_observed_value")
                             and not getattr(var, "has_observed_value", False) and not getattr(var, "has_random_dataset", False)
                             and torch.is_tensor(val)
                             and type(var.distribution).__name__ not in ("EmpiricalDistribution", "DeterministicDistribution")}
        return res

    obs_model._get_sample = capture_obs

    for est_name, est in (("pathwise", ge.PathwiseDerivativeEstimator), ("blackbox", ge.BlackBoxEstimator)):
        for root in roots.values():
            root.link.parameter.grad = None
        torch.manual_seed(seed)
        np.random.seed(seed)
        with DrawRecorder() as rec:
            loss = inference.ReverseKL(gradient_estimator=est).compute_loss(model, q, None, N)
        loss.backward()
        for k, v in captured.get("minibatch", {}).items():
            if "minibatch/" + k in out:
                assert np.array_equal(out["minibatch/" + k], v), "estimators drew different minibatches"
            out["minibatch/" + k] = v
        for k, v in captured.get("drawn", {}).items():
            if "drawn/" + k in out:
                assert np.array_equal(out["drawn/" + k], v), "estimators drew different values of an observed-by-flag variable"
            out["drawn/" + k] = v
        z = captured["z"]
        noise = match_noise(q, z, rec.draws)
        if est_name == "pathwise":
            for k, v in noise.items():
                out["noise/" + k] = v
            for var, s in z.items():
                if type(var).__name__ != "RootVariable":
                    out["z/" + var.name] = s.detach().numpy().copy()
            np.random.seed(seed)
            if captured.get("drawn"):
                # (a variable observed by flag only is drawn from torch's generator: a second call would draw ANOTHER value — the
                #  per-sample terms are recorded on the draw of the compute_loss call above)
                emp = dict(captured["emp"])
            else:
                emp = model.observed_submodel._get_sample(1, observed=True, differentiable=False)
            zz = dict(z)
            zz.update(emp)
            lp = model.get_p_log_probabilities_from_q_samples(q_samples=zz, empirical_samples=emp,
                                                              for_gradient=True, q_model=q)
            H = q._get_entropy(zz, for_gradient=True)
            lq = q.calculate_log_probability(zz)
            out["lp"] = lp.detach().numpy().copy()
            out["H"] = H.detach().numpy().copy()
            out["lq"] = lq.detach().numpy().copy()
        else:
            for k, v in noise.items():
                assert np.array_equal(out["noise/" + k], v), "estimators drew different noise"
        out["loss_" + est_name] = np.float32(loss.detach().numpy())
        for pname, root in roots.items():
            g = root.link.parameter.grad
            out["grad_%s/%s" % (est_name, pname)] = (np.zeros_like(out["param/" + pname]) if g is None
                                                     else g.detach().numpy().copy())
            out["gradnone_%s/%s" % (est_name, pname)] = np.array(g is None)

    if CASES[name][0] in TAYLOR1_BUILDERS:
        # third estimator of the reference on the same draws (gradient_estimators.py:47-56)
        for root in roots.values():
            root.link.parameter.grad = None
        torch.manual_seed(seed)
        np.random.seed(seed)
        with DrawRecorder() as rec:
            loss = inference.ReverseKL(gradient_estimator=ge.Taylor1Estimator).compute_loss(model, q, None, N)
        loss.backward()
        noise = match_noise(q, captured["z"], rec.draws)
        for k, v in noise.items():
            assert np.array_equal(out["noise/" + k], v), "estimators drew different noise"
        out["loss_taylor1"] = np.float32(loss.detach().numpy())
        for pname, root in roots.items():
            g = root.link.parameter.grad
            out["grad_taylor1/" + pname] = np.zeros_like(out["param/" + pname]) if g is None else g.detach().numpy().copy()

    if CASES[name][0] in CUSTOM_ESTIMATOR_BUILDERS:
        # user-defined estimators (the GradientEstimator seam, gradient_estimators.py:17-26) on the same draws
        import brancher_amd.workloads as Wm
        for est_name, est in sorted(Wm.custom_estimators(ge).items()):
            for root in roots.values():
                root.link.parameter.grad = None
            torch.manual_seed(seed)
            np.random.seed(seed)
            with DrawRecorder() as rec:
                loss = inference.ReverseKL(gradient_estimator=est).compute_loss(model, q, None, N)
            loss.backward()
            noise = match_noise(q, captured["z"], rec.draws)
            for k, v in noise.items():
                assert np.array_equal(out["noise/" + k], v), "estimators drew different noise"
            out["loss_custom_" + est_name] = np.float32(loss.detach().numpy())
            for pname, root in roots.items():
                g = root.link.parameter.grad
                out["grad_custom_%s/%s" % (est_name, pname)] = (np.zeros_like(out["param/" + pname]) if g is None
                                                                else g.detach().numpy().copy())

    if all(type(v).__name__ == "RootVariable" for v in q.flatten()):
        # point estimates: the reference's MAP inference method on the same model (inference.py:251-275)
        out["loss_map"] = np.float32(inference.MAP().compute_loss(model, q, None, 1).detach().numpy().reshape(-1)[0])

    if traj is not None:
        # the optimisation loop of inference.py:77-108 (harness; one loss per iteration)
        method = inference.ReverseKL(gradient_estimator=ge.PathwiseDerivativeEstimator)
        opt_kwargs = {k: v for k, v in traj.items() if k not in ("iters", "n", "optimizer")}
        opts = []
        for m in (q, model):
            o = ProbabilisticOptimizer(m, traj["optimizer"], **opt_kwargs)
            if o.optimizer:
                opts.append(o)
        torch.manual_seed(seed + 1000)
        np.random.seed(seed + 1000)
        losses, noise_seq, mb_seq, drawn_seq = [], {}, {}, {}
        for it in range(traj["iters"]):
            with DrawRecorder() as rec:
                loss = method.compute_loss(model, q, None, traj["n"])
            noise = match_noise(q, captured["z"], rec.draws)
            for k, v in noise.items():
                noise_seq.setdefault(k, []).append(v)
            for k, v in captured.get("minibatch", {}).items():
                mb_seq.setdefault(k, []).append(v)
            for k, v in captured.get("drawn", {}).items():
                drawn_seq.setdefault(k, []).append(v)
            if torch.isfinite(loss.detach()).all().item():
                [o.zero_grad() for o in opts]
                loss.backward()
                opts[0].update()
                if it > 0:
                    [o.update() for o in opts[1:]]
            losses.append(float(loss.detach()))
        out["traj/losses"] = np.array(losses, dtype=np.float32)
        for k, v in noise_seq.items():
            out["traj/noise/" + k] = np.stack(v)
        for k, v in mb_seq.items():
            out["traj/minibatch/" + k] = np.stack(v)
        for k, v in drawn_seq.items():
            out["traj/drawn/" + k] = np.stack(v)
        for pname, root in roots.items():
            out["traj/param_after/" + pname] = root.value.detach().numpy().copy()

    meta = dict(case=name, builder=builder, kwargs=kwargs, N=N, seed=seed, trajectory=traj,
                torch=torch.__version__, numpy=np.__version__, posterior_order="sorted by name",
                reference="LucaAmbrogioni/Brancher @ /root/reference")
    out["meta"] = np.array(json.dumps(meta))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-28s loss_pathwise=%.6f loss_blackbox=%.6f  (%d arrays)" % (
        name, out["loss_pathwise"], out["loss_blackbox"], len(out)))


if __name__ == "__main__":
    api = reference_api()
    todo = sys.argv[1:] or list(CASES)
    for case in todo:
        run_case(case, api)
