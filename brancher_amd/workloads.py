"""
Workload definitions (the reference's `examples/` and README models, BASELINE.json configs).

Every builder takes an ``api`` namespace providing the Brancher constructor names
(``NormalVariable``, ``ProbabilisticModel``, ``BF`` ...).  The same builder is therefore
used three ways: with this package (tests, bench), with the oracle, and — inside this
container only — with the real reference imported from /root/reference to generate the
golden vectors (`oracle/gen_golden.py`).
"""
import types

import numpy as np


def native_api():
    from brancher_amd import standard_variables as sv, variables as v, functions as BF
    return types.SimpleNamespace(
        NormalVariable=sv.NormalVariable, LogitNormalVariable=sv.LogitNormalVariable, LogNormalVariable=sv.LogNormalVariable, BetaVariable=sv.BetaVariable,
        BinomialVariable=sv.BinomialVariable, BernulliVariable=sv.BernulliVariable,
        CauchyVariable=sv.CauchyVariable, LaplaceVariable=sv.LaplaceVariable,
        DeterministicVariable=sv.DeterministicVariable, RootVariable=v.RootVariable,
        CategoricalVariable=sv.CategoricalVariable, EmpiricalVariable=sv.EmpiricalVariable,
        RandomIndices=sv.RandomIndices, MultivariateNormalVariable=sv.MultivariateNormalVariable,
        ProbabilisticModel=v.ProbabilisticModel, BF=BF, name="brancher_amd")


def custom_estimators(ge):
    """User-defined gradient estimators written against the reference's seam (`gradient_estimators.py:17-26`: ctor
    `(function, sampler, empirical_samples)`, `__call__(n_samples)` -> scalar).  `ge` is the gradient_estimators module of
    either library: the SAME class bodies run on the reference (PyTorch-CPU autograd) and here (two passes of the fused
    kernel, engine.custom_estimator_loss)."""
    import torch

    class BaselineEstimator(ge.GradientEstimator):
        """score-function estimator with the batch mean of f as a baseline (a control variate), plus the pathwise term"""

        def __call__(self, n_samples):
            samples = self.sampler._get_sample(n_samples, differentiable=False)
            samples.update(self.empirical_samples)
            f = self.function(samples)
            log_q = self.sampler.calculate_log_probability(samples)
            return (log_q * (f - f.mean()).detach() + f).mean()

    class SoftmaxWeightedEstimator(ge.GradientEstimator):
        """pathwise gradients weighted by a softmax of the (detached) values: favours the better samples"""

        def __call__(self, n_samples):
            samples = self.sampler._get_sample(n_samples, differentiable=True)
            samples.update(self.empirical_samples)
            f = self.function(samples)
            w = torch.softmax(0.1 * f.detach().reshape(-1), dim=0).reshape(f.shape)
            return (w * f).sum()

    return dict(baseline=BaselineEstimator, softmax=SoftmaxWeightedEstimator)


def ar_data(T, seed=0, b=0.8, driving_noise=1.0, measure_noise=0.3):
    rng = np.random.RandomState(seed)
    x = np.zeros(T)
    x[0] = rng.normal(0., driving_noise)
    for t in range(1, T):
        x[t] = b * x[t - 1] + rng.normal(0., driving_noise)
    return (x + rng.normal(0., measure_noise, size=T)).astype(np.float32)


def build_readme_ar(api, T=20, data=None, driving_noise=1., measure_noise=0.3, logit_normal=False):
    """BASELINE config 1/3: the README state-space AR model (`README.md:25-75`) with the
    logit-Normal coefficient written as a Normal latent + sigmoid, distinct
    variable names for x and y (the README's duplicate 'x0' is a typo).
    logit_normal=True writes the coefficient the way the README does — ``b = LogitNormalVariable(0.5, 1., ...)`` and
    ``b * x[t-1]`` — with this package's LogitNormalVariable (the reference snapshot has none): the same graph."""
    BF = api.BF
    if data is None:
        data = ar_data(T, driving_noise=driving_noise, measure_noise=measure_noise)
    x0 = api.NormalVariable(0., driving_noise, 'x0')
    y0 = api.NormalVariable(x0, measure_noise, 'y0')
    bl = api.LogitNormalVariable(0.5, 1., 'b_logit') if logit_normal else api.NormalVariable(0.5, 1., 'b_logit')
    coefficient = (lambda v: bl * v) if logit_normal else (lambda v: BF.sigmoid(bl) * v)
    x, y = [x0], [y0]
    for t in range(1, T):
        x.append(api.NormalVariable(coefficient(x[t - 1]), driving_noise, "x{}".format(t)))
        y.append(api.NormalVariable(x[t], measure_noise, "y{}".format(t)))
    model = api.ProbabilisticModel(x + y)
    for t, yt in enumerate(y):
        yt.observe(np.array([[float(data[t])]], dtype=np.float32))

    Qb = (api.LogitNormalVariable if logit_normal else api.NormalVariable)(0.5, 0.5, "b_logit", learnable=True)
    logit_b_post = api.DeterministicVariable(0., 'logit_b_post', learnable=True)
    Qx = [api.NormalVariable(0., 1., 'x0', learnable=True)]
    Qx_mean = [api.DeterministicVariable(0., 'x0_mean', learnable=True)]
    for t in range(1, T):
        Qx_mean.append(api.DeterministicVariable(0., "x{}_mean".format(t), learnable=True))
        Qx.append(api.NormalVariable(BF.sigmoid(logit_b_post) * Qx[t - 1] + Qx_mean[t], 1., "x{}".format(t),
                                     learnable=True))
    posterior = api.ProbabilisticModel([Qb] + Qx)
    model.set_posterior_model(posterior)
    return model


def build_beta_ar(api, T=20, data=None, driving_noise=1., measure_noise=0.5):
    """`tests/test_advanced_autoregressive.py:11-55` style: Beta coefficient, RootVariable
    posterior means, structured Normal chain."""
    if data is None:
        data = ar_data(T, driving_noise=driving_noise, measure_noise=measure_noise)
    x0 = api.NormalVariable(0., driving_noise, 'x0')
    y0 = api.NormalVariable(x0, measure_noise, 'y0')
    b = api.BetaVariable(1., 1., 'b')
    x, y = [x0], [y0]
    for t in range(1, T):
        x.append(api.NormalVariable(b * x[t - 1], driving_noise, "x{}".format(t)))
        y.append(api.NormalVariable(x[t], measure_noise, "y{}".format(t)))
    model = api.ProbabilisticModel(x + y)
    for t, yt in enumerate(y):
        yt.observe(np.array([[float(data[t])]], dtype=np.float32))
    Qb = api.BetaVariable(1., 1., "b", learnable=True)
    coupling = api.RootVariable(0., 'logit_b_post', learnable=True)
    Qx = [api.NormalVariable(0., 1., 'x0', learnable=True)]
    Qx_mean = [api.RootVariable(0., 'x0_mean', learnable=True)]
    for t in range(1, T):
        Qx_mean.append(api.RootVariable(0., "x{}_mean".format(t), learnable=True))
        Qx.append(api.NormalVariable(coupling * Qx[t - 1] + Qx_mean[t], 1., "x{}".format(t), learnable=True))
    model.set_posterior_model(api.ProbabilisticModel([Qb] + Qx))
    return model


def build_beta_binomial(api, n_obs=30, p_real=0.8, seed=0, number_tosses=1):
    """BASELINE config 2: `examples/beta_binomial.py:10-24`."""
    rng = np.random.RandomState(seed)
    k_data = rng.binomial(number_tosses, p_real, size=n_obs).astype(np.float32)
    p = api.BetaVariable(1., 1., "p")
    k = api.BinomialVariable(number_tosses, probs=p, name="k")
    model = api.ProbabilisticModel([k, p])
    k.observe(k_data)
    Qp = api.BetaVariable(1., 1., "p", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qp]))
    return model


def build_lognormal_normal(api, n_obs=20, seed=0):
    """`examples/logNormal_normal.py:12-31`: Normal likelihood with LogNormal scale."""
    rng = np.random.RandomState(seed)
    data = rng.normal(-2., 1., size=n_obs).astype(np.float32)
    nu = api.LogNormalVariable(0., 1., "nu")
    mu = api.NormalVariable(0., 10., "mu")
    x = api.NormalVariable(mu, nu, "x")
    model = api.ProbabilisticModel([x])
    x.observe(data)
    Qnu = api.LogNormalVariable(0., 1., "nu", learnable=True)
    Qmu = api.NormalVariable(0., 1., "mu", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qmu, Qnu]))
    return model


def build_observed_ar(api, T=200, seed=0, q_concentration=0.5):
    """`examples/autoregressive.py:11-41`: a fully observed AR(1) chain; the latents are its coefficient
    b ~ Beta and its noise scale nu ~ LogNormal, with a mean-field Beta / LogNormal posterior.
    (The example initialises Qb = Beta(0.5, 0.5); its samples pile up at 0 and 1, where the fp32
    reparameterisation gradient of the reference itself is 1e-3 away from the fp64 value — the parity
    fixture uses q_concentration=2.)"""
    rng = np.random.RandomState(seed)
    series = np.zeros(T, dtype=np.float32)
    series[0] = rng.normal(0., 1.)
    for t in range(1, T):
        series[t] = 0.7 * series[t - 1] + rng.normal(0., 0.6)
    nu = api.LogNormalVariable(0.3, 1., "nu")
    x0 = api.NormalVariable(0., 1., "x0")
    b = api.BetaVariable(0.5, 1.5, "b")
    x = [x0]
    for t in range(1, T):
        x.append(api.NormalVariable(b * x[t - 1], nu, "x{}".format(t)))
    model = api.ProbabilisticModel(x)
    for t, xt in enumerate(x):
        xt.observe(np.array([series[t]], dtype=np.float32))
    Qnu = api.LogNormalVariable(0.5, 1., "nu", learnable=True)
    Qb = api.BetaVariable(q_concentration, q_concentration, "b", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qb, Qnu]))
    return model


def build_multivariate_regression(api, n=100, seed=0):
    """`examples/multivariate_regression.py:12-47`: observed regressors enter as
    `DeterministicVariable(data, name, is_observed=True)` (datapoint axis of n rows), four Normal
    weights and a LogNormal noise scale, Normal likelihood over the n datapoints."""
    rng = np.random.RandomState(seed)
    x_range = np.linspace(-1., 1., n)
    x1 = api.DeterministicVariable(np.sin(2 * np.pi * 2 * x_range), name="x1", is_observed=True)
    x2 = api.DeterministicVariable(x_range, name="x2", is_observed=True)
    b = api.NormalVariable(0., 1., name="b")
    w1 = api.NormalVariable(0., 1., name="w1")
    w2 = api.NormalVariable(0., 1., name="w2")
    w12 = api.NormalVariable(0., 1., name="w12")
    nu = api.LogNormalVariable(0.2, 0.5, name="nu")
    mean = b + w1 * x1 + w2 * x2 + w12 * x1 * x2
    y = api.NormalVariable(mean, nu, name="y")
    model = api.ProbabilisticModel([y])
    Qb = api.NormalVariable(0., 1., name="b", learnable=True)
    Qw1 = api.NormalVariable(0., 1., name="w1", learnable=True)
    Qw2 = api.NormalVariable(0., 1., name="w2", learnable=True)
    Qw12 = api.NormalVariable(0., 1., name="w12", learnable=True)
    Qnu = api.LogNormalVariable(0.2, 0.5, name="nu", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qb, Qw1, Qw2, Qw12, Qnu]))
    # data of the example's shape: one draw of the regression at fixed weights plus noise
    truth = 0.3 - 0.8 * np.sin(2 * np.pi * 2 * x_range) + 0.5 * x_range + 1.2 * np.sin(2 * np.pi * 2 * x_range) * x_range
    data = (truth + rng.normal(0., 0.4, size=n)).astype(np.float32)
    y.observe(np.reshape(data, (n, 1, 1)))
    return model


def build_dynamic_causal_model(api, steps=30, seed=0):
    """`examples/DynamicCausalModeling.py:12-56`: two coupled, fully observed Euler–Maruyama chains x_n, y_n driven
    by a pulse input; the latents are the seven rate / coupling / noise parameters (LogNormal and Normal, mean
    field).  Every transition mean is a four-term affine expression of two parents and three latents."""
    rng = np.random.RandomState(seed)
    T = 12.
    dt = T / float(steps)
    time_range = np.linspace(0., T, steps)
    pulse = [1. if (1. < t < 2.) or (7. < t < 8.) else 0. for t in time_range]
    # data: one Euler-Maruyama path at a=1, b=1, c=0, d=3, e=10, xi=chi=0.5
    xs, ys = [float(rng.normal(0., 1.))], [float(rng.normal(0., 1.))]
    for n in range(steps):
        xs.append((1 - dt * 1.) * xs[-1] + dt * 0. * ys[-1] + dt * 10. * pulse[n] + np.sqrt(dt) * 0.5 * rng.normal())
        ys.append((1 - dt * 1.) * ys[-2 + 1] + dt * 3. * xs[-2] + np.sqrt(dt) * 0.5 * rng.normal())
    a = api.LogNormalVariable(0., 1., name="a")
    b = api.LogNormalVariable(0., 1., name="b")
    c = api.NormalVariable(0., 2., name="c")
    d = api.NormalVariable(0., 2., name="d")
    e = api.NormalVariable(0., 10., name="e")
    xi = api.LogNormalVariable(0., 0.1, name="xi")
    chi = api.LogNormalVariable(0., 0.1, name="chi")
    x_series = [api.NormalVariable(0., 1., name="x_0")]
    y_series = [api.NormalVariable(0., 1., name="y_0")]
    for n in range(steps):
        x_new_mean = (1 - dt * a) * x_series[-1] + dt * c * y_series[-1] + dt * e * pulse[n]
        y_new_mean = (1 - dt * b) * y_series[-1] + dt * d * x_series[-1]
        x_series += [api.NormalVariable(x_new_mean, np.sqrt(dt) * xi, name="x_{}".format(n + 1))]
        y_series += [api.NormalVariable(y_new_mean, np.sqrt(dt) * chi, name="y_{}".format(n + 1))]
    model = api.ProbabilisticModel([x_series[-1], y_series[-1]])
    for v, val in zip(x_series, xs):
        v.observe(np.array([val], dtype=np.float32))
    for v, val in zip(y_series, ys):
        v.observe(np.array([val], dtype=np.float32))
    Qa = api.LogNormalVariable(0., 0.5, name="a", learnable=True)
    Qb = api.LogNormalVariable(0., 0.5, name="b", learnable=True)
    Qc = api.NormalVariable(0., 0.1, name="c", learnable=True)
    Qd = api.NormalVariable(0., 0.1, name="d", learnable=True)
    Qe = api.NormalVariable(0., 5., name="e", learnable=True)
    Qxi = api.LogNormalVariable(0.1, 0.1, name="xi", learnable=True)
    Qchi = api.LogNormalVariable(0.1, 0.1, name="chi", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qa, Qb, Qc, Qd, Qe, Qxi, Qchi]))
    return model


def build_gp_regression(api, n=8, length_scale=0.6, jitter=1e-3, noise=0.2, seed=0):
    """Gaussian-process regression at fixed inputs (development_playgrounds/GP_playground.py; `stochastic_processes.py:
    29-40,83-95`): f ~ MultivariateNormal(0, K) with the squared-exponential covariance of the inputs — a constant of
    the model —, y ~ Normal(f, noise) observed, mean-field Normal posterior over the function values."""
    rng = np.random.RandomState(seed)
    x = np.linspace(-2., 2., n)
    K = np.exp(-(x[:, None] - x[None, :]) ** 2 / (2 * length_scale ** 2)) + jitter * np.eye(n)
    f = api.MultivariateNormalVariable(loc=np.zeros((n,)), covariance_matrix=K, name="f")
    y = api.NormalVariable(f, noise, name="y")
    model = api.ProbabilisticModel([y])
    data = np.sin(2 * np.pi * 0.4 * x) + noise * rng.normal(0., 1., (1, n))
    y.observe(data.astype(np.float32))
    Qf = api.NormalVariable(loc=np.zeros((n,)), scale=0.8, name="f", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qf]))
    return model


def build_gp_hyperparameters(api, n=5, jitter=1e-2, noise=0.2, seed=0, learnable_amplitude=True, structured_mean=False):
    """A Gaussian process whose kernel hyper-parameters are inferred: f ~ MultivariateNormal(0, K(ell, amp)) with the
    squared-exponential covariance  K = amp * exp(-sqdist / (2 ell^2)) + jitter I  of fixed inputs, a LogNormal latent
    length-scale `ell` (inferred with a LogNormal posterior) and — type-II maximum likelihood — a learnable amplitude in the
    joint model; y ~ Normal(f, noise) observed (`distributions.py:314-331`, `standard_variables.py:317-347`; the covariance
    is an ordinary link expression, `stochastic_processes.py:29-40` builds it the same way from a kernel function)."""
    BF = api.BF
    rng = np.random.RandomState(seed)
    x = np.linspace(-2., 2., n)
    sqdist = api.RootVariable(((x[:, None] - x[None, :]) ** 2).astype(np.float32), "sqdist")
    eye = api.RootVariable((jitter * np.eye(n)).astype(np.float32), "jitter")
    ell = api.LogNormalVariable(-0.5, 0.3, "ell")
    amp = api.RootVariable(1.3, "amplitude", learnable=True) if learnable_amplitude else 1.3
    K = BF.exp(sqdist * (-0.5) / (ell * ell)) * amp + eye
    f = api.MultivariateNormalVariable(loc=np.zeros((n,)), covariance_matrix=K, name="f")
    if structured_mean:
        # an offset per observation, and a posterior whose mean of f FOLLOWS the sampled offsets: q(f | shift) =
        # Normal(gain * shift + mean, s) — under the Taylor1 estimator the value of the multivariate-normal term is then an
        # expression of a sampled parent (`gradient_estimators.py:47-56`), not a vector of parameters
        shift = api.NormalVariable(np.zeros((n, 1)), 0.5, "shift")
        y = api.NormalVariable(f + shift, noise, name="y")
    else:
        y = api.NormalVariable(f, noise, name="y")
    model = api.ProbabilisticModel([y])
    y.observe((np.sin(2 * np.pi * 0.3 * x) + noise * rng.normal(0., 1., (1, n))).astype(np.float32))
    Qell = api.LogNormalVariable(-0.4, 0.2, "ell", learnable=True)
    if structured_mean:
        Qshift = api.NormalVariable(np.full((n, 1), 0.1), 0.3, "shift", learnable=True)
        gain = api.DeterministicVariable(np.full((n, 1), -0.5), "f_gain", learnable=True)
        mean = api.DeterministicVariable(np.zeros((n, 1)), "f_mean", learnable=True)
        Qf = api.NormalVariable(Qshift * gain + mean, 0.8, name="f", learnable=True)
        model.set_posterior_model(api.ProbabilisticModel([Qell, Qshift, Qf]))
        return model
    Qf = api.NormalVariable(loc=np.zeros((n,)), scale=0.8, name="f", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qell, Qf]))
    return model


def build_gp_marginal_likelihood(api, n=200, noise=0.3, seed=0):
    """Gaussian-process regression with the function values integrated out: y ~ MultivariateNormal(0, K(ell, amp) + noise^2 I)
    OBSERVED, the squared-exponential covariance an elementwise link expression of a LogNormal latent length-scale (LogNormal
    posterior) and a learnable amplitude (`distributions.py:314-331`, `standard_variables.py:317-347`,
    `stochastic_processes.py:29-40`).  The per-sample program is a handful of records; all the work is the n x n factorisation
    per Monte-Carlo sample — the batched kernel's, in LDS up to n = 192 and in device memory beyond."""
    BF = api.BF
    rng = np.random.RandomState(seed)
    x = np.linspace(-3., 3., n)
    sqdist = api.RootVariable(((x[:, None] - x[None, :]) ** 2).astype(np.float32), "sqdist")
    eye = api.RootVariable((noise ** 2 * np.eye(n)).astype(np.float32), "noise")
    ell = api.LogNormalVariable(-0.5, 0.3, "ell")
    amp = api.RootVariable(1.3, "amplitude", learnable=True)
    K = BF.exp(sqdist * (-0.5) / (ell * ell)) * amp + eye
    y = api.MultivariateNormalVariable(loc=np.zeros((n,)), covariance_matrix=K, name="y")
    model = api.ProbabilisticModel([y])
    y.observe((np.sin(2 * np.pi * 0.25 * x) + noise * rng.normal(0., 1., (1, n))).astype(np.float32))
    Qell = api.LogNormalVariable(-0.4, 0.2, "ell", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qell]))
    return model


def build_mvn_forms(api, n=16, form="scale_tril", noise=0.3, seed=0):
    """The other two parameterisations of `MultivariateNormalVariable` (`standard_variables.py:317-347`,
    `distributions.py:314-331`) with a matrix that depends on a sampled scalar, as an ELEMENTWISE link expression of constant
    matrices: `scale_tril = L0 * s + jitter I` (L0 the Cholesky factor of a squared-exponential kernel) or
    `precision_matrix = P0 * tau + jitter I` (P0 its inverse) with a LogNormal latent scale inferred by a LogNormal
    posterior; y ~ Normal(f, noise) observed."""
    rng = np.random.RandomState(seed)
    x = np.linspace(-2., 2., n)
    K0 = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / 0.8 ** 2) + 0.05 * np.eye(n)
    if form == "scale_tril":
        base, jit = np.linalg.cholesky(K0), 0.02
    elif form == "precision_matrix":
        base, jit = np.linalg.inv(K0), 0.05
    else:
        base, jit = K0, 0.02
    mat0 = api.RootVariable(base.astype(np.float32), "base_matrix")
    eye = api.RootVariable((jit * np.eye(n)).astype(np.float32), "jitter")
    s = api.LogNormalVariable(0.1, 0.3, "s")
    f = api.MultivariateNormalVariable(loc=np.zeros((n,)), name="f", **{form: mat0 * s + eye})
    y = api.NormalVariable(f, noise, name="y")
    model = api.ProbabilisticModel([y])
    y.observe((np.sin(2 * np.pi * 0.3 * x) + noise * rng.normal(0., 1., (1, n))).astype(np.float32))
    Qs = api.LogNormalVariable(0.0, 0.2, "s", learnable=True)
    Qf = api.NormalVariable(loc=np.zeros((n,)), scale=0.7, name="f", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qs, Qf]))
    return model


def build_module_link_regression(api, n_obs=6, hidden=4, seed=0, n_in=1, activation="Tanh", hidden2=0, n_out=1):
    """A `torch.nn.Module` as a link on the scalar path (`brancher/functions.py:15-41`): the response is a small MLP of a
    latent, `y_i ~ N(net(z) * x_i + c, 0.4)`, `net = Linear(1, H) -> Tanh -> Linear(H, 1)`, z ~ N(0, 1) with a learnable Normal
    posterior.  The reference calls the module on the sample and steps its tensors with the joint model's optimizer
    (`optimizers.py:36-49`, from iteration 1 on: `inference.py:102-104`); here they are learnable entries of the program."""
    import torch
    BF = api.BF
    gen = torch.Generator().manual_seed(seed + 11)
    act = getattr(torch.nn, activation)
    stages = [torch.nn.Linear(n_in, hidden), act()]
    if hidden2:
        stages += [torch.nn.Linear(hidden, hidden2), act()]
    stages.append(torch.nn.Linear(hidden2 or hidden, n_out))
    net = torch.nn.Sequential(*stages)                 # (n_in > 1: the module acts on a row vector of latents, [1, n_in])
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.7)
    rng = np.random.RandomState(seed)
    xs = rng.uniform(-1., 1., (n_obs, 1, 1)).astype(np.float32)
    data = (0.8 * xs + 0.1 + 0.05 * rng.normal(0., 1., (n_obs, 1, 1))).astype(np.float32)
    if n_in == 1:
        z, qz = api.NormalVariable(0., 1., "z"), api.NormalVariable(0.3, 0.6, "z", learnable=True)
    else:
        z = api.NormalVariable(np.zeros((1, n_in)), np.ones((1, n_in)), "z")
        qz = api.NormalVariable(0.3 * rng.normal(0., 1., (1, n_in)), 0.6 * np.ones((1, n_in)), "z", learnable=True)
    f = BF.BrancherFunction(net, name="net")
    x = api.DeterministicVariable(xs, "x", is_observed=True)              # [datapoints, 1, 1]
    if n_out > 1:
        # (round 6) several output units: the response is a vector per datapoint, y_i ~ N(net(z) * x_i + c, 0.4) with net(z) in R^n_out
        data = (data * np.linspace(0.5, 1.5, n_out).reshape(1, 1, n_out) + 0.05 * rng.normal(0., 1., (n_obs, 1, n_out))).astype(np.float32)
    y = api.NormalVariable(f(z) * x + 0.1, 0.4, "y")
    model = api.ProbabilisticModel([y])
    y.observe(data)
    model.set_posterior_model(api.ProbabilisticModel([qz]))
    model._golden_modules = {"net": net}
    return model


def build_map_estimate(api, n_obs=12, seed=0):
    """Point estimates (MAP, `inference.py:251-275`; `examples/MAP_logistic_regression.py:46-56`): the "posterior" is a
    model of learnable RootVariables carrying the latents' names.  No sampling and no entropy: the loss is
    -log p(theta, data)."""
    rng = np.random.RandomState(seed)
    mu = api.NormalVariable(0., 10., "mu")
    nu = api.LogNormalVariable(0., 1., "nu")
    x = api.NormalVariable(mu, nu, "x")
    model = api.ProbabilisticModel([x])
    x.observe(rng.normal(1.0, 2.0, size=n_obs).astype(np.float32))
    model.set_posterior_model(api.ProbabilisticModel([api.RootVariable(0.3, "mu", learnable=True),
                                                      api.RootVariable(1.5, "nu", learnable=True)]))
    return model


def build_vector_latent(api, n_obs=9, dim=4, seed=0):
    """Vector-valued nodes: a latent z in R^dim with an elementwise non-linear link into an observed x in R^dim
    over n_obs datapoints, and a second vector latent whose scale is a LogNormal scalar (broadcast)."""
    BF = api.BF
    rng = np.random.RandomState(seed)
    data = rng.normal(0.5, 1.0, size=(n_obs, dim, 1)).astype(np.float32)
    s = api.LogNormalVariable(0., 0.3, "s")
    z = api.NormalVariable(np.linspace(-1., 1., dim).reshape(dim, 1), 1.5 * np.ones((dim, 1)), "z")
    u = api.NormalVariable(BF.tanh(z) * 0.5, s, "u")
    x = api.NormalVariable(z * 2. + u, 0.8 * np.ones((dim, 1)), "x")
    model = api.ProbabilisticModel([x])
    x.observe(data)
    Qs = api.LogNormalVariable(0.1, 0.2, "s", learnable=True)
    Qz = api.NormalVariable(np.zeros((dim, 1)), np.ones((dim, 1)), "z", learnable=True)
    Qu = api.NormalVariable(Qz * 0.3, 0.7 * np.ones((dim, 1)), "u", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qs, Qz, Qu]))
    return model


def build_scale_from_latent(api, n_obs=6, seed=0):
    """A posterior whose SCALES are sampled: u ~ Normal(0.3 z, 0.7 s) and w ~ LogNormal(0.2 u, 0.4 s) with s a LogNormal
    latent of q.  Under the Taylor1 estimator (`gradient_estimators.py:47-56`) the entropies of u and w are evaluated on
    the MEANS of s and u, not on their draws."""
    BF = api.BF
    rng = np.random.RandomState(seed)
    data = rng.normal(0.5, 1.0, size=(n_obs, 1)).astype(np.float32)
    s = api.LogNormalVariable(0., 0.3, "s")
    z = api.NormalVariable(0., 1.5, "z")
    u = api.NormalVariable(BF.tanh(z) * 0.5, s, "u")
    w = api.LogNormalVariable(0.1 * u, 0.5, "w")
    x = api.NormalVariable(z * 2. + u, 0.8 * w, "x")
    model = api.ProbabilisticModel([x])
    x.observe(data)
    Qs = api.LogNormalVariable(0.1, 0.2, "s", learnable=True)
    Qz = api.NormalVariable(0., 1., "z", learnable=True)
    Qu = api.NormalVariable(Qz * 0.3, Qs * 0.7, "u")
    Qw = api.LogNormalVariable(Qu * 0.2, Qs * 0.4, "w")
    model.set_posterior_model(api.ProbabilisticModel([Qs, Qz, Qu, Qw]))
    return model


def build_linear_predictor(api, n_obs=5, dim=4, seed=0):
    """Axis reductions and static indexing inside links (`functions.py:50-62` wraps torch.sum / torch.transpose;
    `variables.py:279-289,1037-1053` the `[...]` of variables and links).  Inside a link a value is laid out
    [samples x datapoints, d1, d2] (`variables.py:436-449`), so ``dim=1`` is the first element axis.  A weight vector w in
    R^dim, a regression on n_obs feature rows through ``BF.sum(w * features, dim=1, keepdim=True)``, and a second
    observation built from one element of w (a slice keeps the axis, so the operands of the sum keep one rank) and from a
    row-vector product after ``BF.transpose``, a third from an integer index."""
    BF = api.BF
    rng = np.random.RandomState(seed)
    features = rng.normal(0., 1., size=(n_obs, dim, 1)).astype(np.float32)
    targets = rng.normal(0.3, 1.2, size=(n_obs, 1, 1)).astype(np.float32)
    probe = rng.normal(-0.2, 0.5, size=(3, 1, 1)).astype(np.float32)
    row = np.linspace(0.5, -1.0, dim).reshape(1, dim).astype(np.float32)
    w = api.NormalVariable(np.zeros((dim, 1)), np.ones((dim, 1)), "w")
    b = api.NormalVariable(0., 2., "b")
    feats = api.DeterministicVariable(features, "features", is_observed=True)     # [datapoints, dim, 1]
    rowv = api.RootVariable(row, "row")                                            # [1, dim]: a row vector
    y = api.NormalVariable(BF.sum(w * feats, dim=1, keepdim=True) + b, 0.6, "y")
    t = api.NormalVariable(w[(slice(2, 3),)] * 2. + BF.sum(BF.transpose(w, 1, 2) * rowv, dim=2, keepdim=True), 0.7, "t")
    u = api.NormalVariable(w[1], 0.9, "u")              # an integer index drops the axis: [rows, 1]
    model = api.ProbabilisticModel([y, t, u])
    y.observe(targets)
    t.observe(probe)
    u.observe(probe[:2] + 0.4)
    Qw = api.NormalVariable(0.1 * np.ones((dim, 1)), 0.8 * np.ones((dim, 1)), "w", learnable=True)
    Qb = api.NormalVariable(0.2, 1.1, "b", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qw, Qb]))
    return model


def build_flat_vector_sum(api, n_obs=6, dim=5, seed=0):
    """A latent written as a FLAT array: ``NormalVariable(np.zeros(dim), np.ones(dim), "w")`` stores its parameters
    [1, 1, dim] (`utilities.py:236`), so inside links w is [rows, dim] — rank 2 — and ``dim=-1`` / ``dim=1`` both name its
    dim axis (`functions.py:50-62` hands them to torch.sum as they are).  A weighted sum of w with a flat coefficient
    array through ``BF.sum(w * coef, dim=-1, keepdim=True)`` and an unweighted one through ``dim=1`` feed two observed
    Normal variables."""
    BF = api.BF
    rng = np.random.RandomState(seed)
    coef = api.RootVariable(np.linspace(-1., 1.5, dim).astype(np.float32), "coef")          # [1, 1, dim]
    targets = rng.normal(0.5, 1.0, size=(n_obs,)).astype(np.float32)
    w = api.NormalVariable(np.zeros(dim), np.ones(dim), "w")
    y = api.NormalVariable(BF.sum(w * coef, dim=-1, keepdim=True), 0.6, "y")
    s = api.NormalVariable(BF.sum(w, dim=1, keepdim=True) * 0.5, 0.9, "s")
    model = api.ProbabilisticModel([y, s])
    y.observe(targets)
    s.observe(targets[:3] - 0.2)
    Qw = api.NormalVariable(0.1 * np.ones(dim), 0.7 * np.ones(dim), "w", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qw]))
    return model


def build_softmax_classifier(api, n_obs=6, n_classes=3, seed=0):
    """An observed CategoricalVariable whose logits are an elementwise link (no matmul): per-class slopes and offsets
    (two vector latents) of one scalar regressor, labels observed (`standard_variables.py:280-299`,
    `distributions.py:275-311`)."""
    BF = api.BF
    rng = np.random.RandomState(seed)
    regressor = rng.normal(0., 1.2, size=(n_obs, 1, 1)).astype(np.float32)
    labels = rng.randint(0, n_classes, size=(n_obs, 1)).astype(np.float32)
    x = api.DeterministicVariable(regressor, "regressor", is_observed=True)
    slope = api.NormalVariable(np.zeros((n_classes, 1)), np.ones((n_classes, 1)), "slope")
    offset = api.NormalVariable(np.zeros((n_classes, 1)), 2. * np.ones((n_classes, 1)), "offset")
    k = api.CategoricalVariable(logits=slope * x + BF.tanh(offset), name="k")
    model = api.ProbabilisticModel([k])
    k.observe(labels)
    Qslope = api.NormalVariable(0.1 * np.ones((n_classes, 1)), 0.7 * np.ones((n_classes, 1)), "slope", learnable=True)
    Qoffset = api.NormalVariable(np.linspace(-0.3, 0.3, n_classes).reshape(n_classes, 1), 0.9 * np.ones((n_classes, 1)),
                                 "offset", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qslope, Qoffset]))
    return model


def build_learnable_model(api, n_obs=15, seed=0):
    """Learnable parameters in the JOINT model as well as in the posterior (type-II maximum likelihood): the
    likelihood's scale and the prior's location are `learnable=True` roots of p.  `perform_inference` then runs two
    optimizers — the posterior's always, the model's only when `iteration > pretraining_iterations`
    (`inference.py:77-88,102-104`)."""
    rng = np.random.RandomState(seed)
    data = rng.normal(0.8, 1.3, size=n_obs).astype(np.float32)
    # (an explicitly named root: the auto-created roots of a prior variable `mu` would be called mu_loc / mu_scale
    #  like the posterior's, and the reference then silently substitutes the posterior's — DESIGN.md §2)
    prior_loc = api.RootVariable(0.2, "prior_loc", learnable=True)  # learnable prior location
    mu = api.NormalVariable(prior_loc, 2., "mu")
    x = api.NormalVariable(mu, 1., "x", learnable=True)             # learnable likelihood scale (root x_scale)
    model = api.ProbabilisticModel([x])
    x.observe(data)
    Qmu = api.NormalVariable(0., 1., "mu", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qmu]))
    return model


def build_discrete_latent(api, n_obs=8, seed=0):
    """A discrete (non-reparameterisable) latent: z ~ Bernoulli shifts the mean of a Normal likelihood, q(z) is a
    Bernoulli with a learnable logit.  Its gradient exists only through the score-function term of the BlackBox
    estimator (`gradient_estimators.py:31-36`; `.sample()` at distributions.py:123-124); the Pathwise estimator
    gives it none, exactly as in the reference."""
    rng = np.random.RandomState(seed)
    z = api.BernulliVariable(probs=0.3, name="z")
    m = api.NormalVariable(0., 1., "m")
    x = api.NormalVariable(m + z * 2.0, 0.7, "x")
    model = api.ProbabilisticModel([x])
    x.observe(rng.normal(1.5, 0.7, size=n_obs).astype(np.float32))
    Qz = api.BernulliVariable(logits=0.2, name="z", learnable=True)
    Qm = api.NormalVariable(0.1, 0.8, "m", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qz, Qm]))
    return model


def build_heavy_tails(api, n_obs=12, seed=1):
    """Cauchy / Laplace coverage (`examples/logNormal_normal.py` imports both): a Laplace
    location with Cauchy likelihood and an explicit nonlinear link."""
    BF = api.BF
    rng = np.random.RandomState(seed)
    data = rng.standard_cauchy(size=n_obs).astype(np.float32) * 0.5 + 1.0
    m = api.LaplaceVariable(0., 2., "m")
    s = api.LogNormalVariable(0., 0.5, "s")
    x = api.CauchyVariable(BF.tanh(m) * 2., s + 0.1, "x")
    model = api.ProbabilisticModel([x])
    x.observe(data)
    Qm = api.LaplaceVariable(0.3, 1., "m", learnable=True)
    Qs = api.LogNormalVariable(0.1, 0.4, "s", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qm, Qs]))
    return model


def logreg_data(dataset_size, n_features, n_classes, seed=0, pixels="unit"):
    """Synthetic stand-in for MNIST (not available offline; SURVEY §8d cfg 4) with uniformly random labels.
    pixels="uint8": pixel counts 0..255 as floats — what `examples/MNIST_logistic_regression.py:15-19` feeds the model
    (`train.train_data.numpy()`, no normalisation) and what SURVEY §8d prescribes; pixels="unit": features in [0, 1]."""
    rng = np.random.RandomState(seed)
    if pixels == "uint8":
        X = rng.randint(0, 256, size=(dataset_size, n_features, 1)).astype(np.float32)
    else:
        X = rng.uniform(0., 1., size=(dataset_size, n_features, 1)).astype(np.float32)
    labels = rng.randint(0, n_classes, size=dataset_size)
    return X, labels


def build_binary_logistic_regression(api, dataset_size=50, batch_size=30, n_features=2, seed=0):
    """`examples/minibatch_logistic_regression.py:13-43`: two Gaussian clouds, `BinomialVariable(1, logits =
    matmul(weights, x))` observed through an `EmpiricalVariable` of labels that shares the minibatch indices of x."""
    BF = api.BF
    rng = np.random.RandomState(seed)
    half = dataset_size // 2
    x1 = rng.normal(1.5, 1.5, (half, n_features, 1))
    x2 = rng.normal(-1.5, 1.5, (dataset_size - half, n_features, 1))
    input_variable = np.concatenate((x1, x2), axis=0).astype(np.float32)
    output_labels = np.concatenate((np.zeros((half, 1)), np.ones((dataset_size - half, 1))), axis=0).astype(np.float32)
    indices = api.RandomIndices(dataset_size=dataset_size, batch_size=batch_size, name="indices", is_observed=True)
    x = api.EmpiricalVariable(input_variable, indices=indices, name="x", is_observed=True)
    labels = api.EmpiricalVariable(output_labels, indices=indices, name="labels", is_observed=True)
    weights = api.NormalVariable(np.zeros((1, n_features)), 0.5 * np.ones((1, n_features)), "weights")
    k = api.BinomialVariable(1, logits=BF.matmul(weights, x), name="k")
    model = api.ProbabilisticModel([k])
    k.observe(labels)
    Qweights = api.NormalVariable(np.zeros((1, n_features)), np.ones((1, n_features)), "weights", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qweights]))
    return model


def build_map_logistic_regression(api, dataset_size=30, n_features=4, n_classes=3, seed=0):
    """`examples/MAP_logistic_regression.py:17-56`: multinomial logistic regression whose "posterior" is a learnable
    RootVariable named like the weight matrix — a point estimate trained with `inference.MAP` on the full batch."""
    BF = api.BF
    X, labels = logreg_data(dataset_size, n_features, n_classes, seed)
    X = (X * 6.0).astype(np.float32)                       # iris-like magnitudes
    indices = api.RandomIndices(dataset_size=dataset_size, batch_size=dataset_size, name="indices", is_observed=True)
    x = api.EmpiricalVariable(X.reshape(dataset_size, n_features, 1), indices=indices, name="x", is_observed=True)
    lab = api.EmpiricalVariable(labels.astype(np.int32), indices=indices, name="labels", is_observed=True)
    weights = api.NormalVariable(np.zeros((n_classes, n_features)), 10 * np.ones((n_classes, n_features)), "weights")
    k = api.CategoricalVariable(logits=BF.matmul(weights, x), name="k")
    model = api.ProbabilisticModel([k])
    k.observe(lab)
    init = np.random.RandomState(seed + 1).normal(0., 1., (n_classes, n_features))
    model.set_posterior_model(api.ProbabilisticModel([api.RootVariable(init, name="weights", learnable=True)]))
    return model


def build_logistic_regression(api, dataset_size=64, batch_size=32, n_features=784, n_classes=10, seed=0,
                              prior_scale=10., q_scale=0.1, pixels="unit"):
    """BASELINE config 4: Bayesian multinomial logistic regression with a dense `matmul` link and a
    random minibatch per iteration (`examples/MNIST_logistic_regression.py:15-54`)."""
    BF = api.BF
    X, labels = logreg_data(dataset_size, n_features, n_classes, seed, pixels)
    minibatch_indices = api.RandomIndices(dataset_size=dataset_size, batch_size=batch_size, name="indices",
                                          is_observed=True)
    x = api.EmpiricalVariable(X, indices=minibatch_indices, name="x", is_observed=True)
    y = api.EmpiricalVariable(labels, indices=minibatch_indices, name="labels", is_observed=True)
    weights = api.NormalVariable(np.zeros((n_classes, n_features)), prior_scale * np.ones((n_classes, n_features)),
                                 "weights")
    k = api.CategoricalVariable(logits=BF.matmul(weights, x), name="k")
    model = api.ProbabilisticModel([k])
    k.observe(y)
    Qweights = api.NormalVariable(np.zeros((n_classes, n_features)), q_scale * np.ones((n_classes, n_features)),
                                  "weights", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qweights]))
    return model


def build_population_receptive_fields(api, field=40, n_data=15, seed=0):
    """`examples/PopulationReceptiveFields.py:15-45`: the response of a Gaussian receptive field (centre mu_x, mu_y and width v latent)
    to `n_data` stimulus images on a `field` x `field` mesh, `mean_response = BF.sum(BF.sum(receptive_field * input, dim=1), dim=2)` —
    a reduction over field^2 elements per datapoint and Monte-Carlo sample.  The stimulus node `input` is observed BY FLAG only
    (`is_observed=True`, never given a value): the reference draws it from its Normal once per evaluation (`variables.py:849`
    takes `observed_submodel._get_sample(1, observed=True)`, `:553-565` draws what has no value) and all samples share the draw.
    The example samples its data from the model; here the same generative process at the example's true values (mu = (1, 2),
    v = 0.3, nu = 0.1) with numpy."""
    BF = api.BF
    S = 6.
    x_range = np.linspace(-S / 2., S / 2., field)
    x_mesh, y_mesh = np.meshgrid(x_range, x_range)
    x = api.RootVariable(x_mesh, name="x")
    y = api.RootVariable(y_mesh, name="y")
    w1 = api.NormalVariable(0., 1., name="w1")
    w2 = api.NormalVariable(0., 1., name="w2")
    b = api.NormalVariable(0., 1., name="b")
    experimental_input = api.NormalVariable(BF.exp(BF.sin(w1 * x + w2 * y + b)), 0.1, name="input", is_observed=True)
    mu_x = api.NormalVariable(0., 1., name="mu_x")
    mu_y = api.NormalVariable(0., 1., name="mu_y")
    v = api.LogNormalVariable(0., 0.1, name="v")
    nu = api.LogNormalVariable(-1, 0.01, name="nu")
    receptive_field = BF.exp((-(x - mu_x) ** 2 - (y - mu_y) ** 2) / (2. * v ** 2)) / (2. * BF.sqrt(np.pi * v ** 2))
    mean_response = BF.sum(BF.sum(receptive_field * experimental_input, dim=1, keepdim=True), dim=2, keepdim=True)
    response = api.NormalVariable(mean_response, nu, name="response")
    model = api.ProbabilisticModel([response, experimental_input])
    rng = np.random.RandomState(seed)
    w1v, w2v, bv = (rng.normal(0., 1., size=(n_data, 1)) for _ in range(3))
    stim = np.exp(np.sin(w1v[:, :, None] * x_mesh + w2v[:, :, None] * y_mesh + bv[:, :, None])) + 0.1 * rng.normal(size=(n_data, field, field))
    rf = np.exp((-(x_mesh - 1.) ** 2 - (y_mesh - 2.) ** 2) / (2. * 0.3 ** 2)) / (2. * np.sqrt(np.pi * 0.3 ** 2))
    resp = (rf * stim).sum(axis=(1, 2)).reshape(n_data, 1, 1) + 0.1 * rng.normal(size=(n_data, 1, 1))
    w1.observe(w1v.astype(np.float32))
    w2.observe(w2v.astype(np.float32))
    b.observe(bv.astype(np.float32))
    response.observe(resp.astype(np.float32))
    Qmu_x = api.NormalVariable(0., 1., name="mu_x", learnable=True)
    Qmu_y = api.NormalVariable(0., 1., name="mu_y", learnable=True)
    Qv = api.LogNormalVariable(0., 0.1, name="v", learnable=True)
    Qnu = api.LogNormalVariable(-1, 0.01, name="nu", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qmu_x, Qmu_y, Qv, Qnu]))
    return model


def build_minibatch_normal_mean(api, dataset_size=40, batch_size=8, seed=0, own_draw=False):
    """The minibatch data path OUTSIDE the matmul patterns (`standard_variables.py:71-112`): a Normal mean whose
    observations are `batch_size` rows of a dataset, other rows in every evaluation — `y.observe(EmpiricalVariable(...))`,
    `variables.py:572-590`.  `own_draw`: the EmpiricalVariable draws its own rows (`batch_size=`, `distributions.py:436-441`)
    instead of sharing a RandomIndices variable."""
    rng = np.random.RandomState(seed)
    data = rng.normal(1.5, 0.7, size=(dataset_size, 1)).astype(np.float32)
    if own_draw:
        ydata = api.EmpiricalVariable(data, batch_size=batch_size, name="ydata", is_observed=True)
    else:
        indices = api.RandomIndices(dataset_size=dataset_size, batch_size=batch_size, name="indices", is_observed=True)
        ydata = api.EmpiricalVariable(data, indices=indices, name="ydata", is_observed=True)
    mu = api.NormalVariable(0., 10., "mu")
    nu = api.LogNormalVariable(0., 0.5, "nu")                     # the observation noise is latent too
    y = api.NormalVariable(mu, nu, "y")
    model = api.ProbabilisticModel([y])
    y.observe(ydata)
    Qmu = api.NormalVariable(0.3, 1.2, "mu", learnable=True)
    Qnu = api.LogNormalVariable(-0.2, 0.3, "nu", learnable=True)
    model.set_posterior_model(api.ProbabilisticModel([Qmu, Qnu]))
    return model


def build_minibatch_linear_regression(api, dataset_size=40, batch_size=8, n_features=3, n_outputs=1, seed=0, prior="normal",
                                      latent_scale=False):
    """Minibatched Bayesian linear regression, the Normal-likelihood neighbour of `examples/minibatch_logistic_regression.py:13-51`:
    `y ~ Normal(BF.matmul(weights, x), 0.3)` with x and the targets two EmpiricalVariables that share a RandomIndices variable.
    `prior`: "normal" | "laplace" | "cauchy" — the weights' prior; `latent_scale`: the prior's scale is itself a latent
    (`NormalVariable(0, tau, "weights")` with tau ~ LogNormal shared by every weight)."""
    BF = api.BF
    rng = np.random.RandomState(seed)
    X = rng.normal(0., 1., size=(dataset_size, n_features, 1)).astype(np.float32)
    true_w = rng.normal(0., 1.5, size=(n_outputs, n_features))
    Y = (np.einsum("op,dpi->doi", true_w, X) + 0.3 * rng.normal(size=(dataset_size, n_outputs, 1))).astype(np.float32)
    indices = api.RandomIndices(dataset_size=dataset_size, batch_size=batch_size, name="indices", is_observed=True)
    x = api.EmpiricalVariable(X, indices=indices, name="x", is_observed=True)
    targets = api.EmpiricalVariable(Y, indices=indices, name="targets", is_observed=True)
    shape = (n_outputs, n_features)
    q_vars = []
    if latent_scale:
        tau = api.LogNormalVariable(0., 0.5, "tau")
        scale = tau
        q_vars.append(api.LogNormalVariable(0.1, 0.25, "tau", learnable=True))
    else:
        scale = 2.0 * np.ones(shape)
    make = {"normal": api.NormalVariable, "laplace": api.LaplaceVariable, "cauchy": api.CauchyVariable}[prior]
    weights = make(np.zeros(shape), scale, "weights")
    y = api.NormalVariable(BF.matmul(weights, x), 0.3, "y")
    model = api.ProbabilisticModel([y])
    y.observe(targets)
    q_vars.append(api.NormalVariable(0.1 * np.ones(shape), 0.5 * np.ones(shape), "weights", learnable=True))
    model.set_posterior_model(api.ProbabilisticModel(q_vars))
    return model


def build_bayesian_neural_network(api, dataset_size=48, batch_size=30, n_features=784, n_hidden=20, n_classes=10, seed=0,
                                  prior_scale=10., q_scale=0.2, q_scale1=None, q_loc_scale=0.0, pixels="uint8",
                                  activation="tanh", hidden2=0):
    """`tests/test_MNIST_bayesian_neural_network.py:20-60` of the reference: a one-hidden-layer network whose weight matrices AND
    biases are latent — weights1 [H, P], b1 [H, 1], weights2 [C, H], b2 [C, 1], priors N(0, 10), a mean-field Normal posterior
    over all four (scale 0.2) — `tanh(matmul(weights1, x) + b1)`, `matmul(weights2, hidden) + b2` as the logits of an observed
    Categorical, a random minibatch per iteration.  (The reference's file spells the likelihood `softmax_p=`, which its
    CategoricalVariable does not accept; `logits=` is what the sibling examples use.)  MNIST is not available offline: pixel
    counts 0..255 as in `logreg_data`.  q_scale1: the posterior scale of weights1 alone (with raw pixel counts and 0.2 every
    tanh saturates and every gradient of the first layer vanishes: the fixtures use a scale that keeps the units alive);
    q_loc_scale: posterior means drawn at that many posterior scales from zero; hidden2 > 0: a second hidden layer."""
    BF = api.BF
    X, labels = logreg_data(dataset_size, n_features, max(n_classes, 2), seed, pixels)
    if n_classes == 1:
        labels = labels.astype(np.float32).reshape(-1, 1)          # (0 / 1, as minibatch_logistic_regression.py:20 stores them)
    rng = np.random.RandomState(seed + 7)
    act = getattr(BF, activation)
    indices = api.RandomIndices(dataset_size=dataset_size, batch_size=batch_size, name="indices", is_observed=True)
    x = api.EmpiricalVariable(X, indices=indices, name="x", is_observed=True)
    y = api.EmpiricalVariable(labels, indices=indices, name="labels", is_observed=True)
    widths = [n_features, n_hidden] + ([hidden2] if hidden2 else []) + [n_classes]
    layer, q_vars = x, []
    for l in range(1, len(widths)):
        rows, cols = widths[l], widths[l - 1]
        b = api.NormalVariable(np.zeros((rows, 1)), prior_scale * np.ones((rows, 1)), "b%d" % l)
        w = api.NormalVariable(np.zeros((rows, cols)), prior_scale * np.ones((rows, cols)), "weights%d" % l)
        pre = BF.matmul(w, layer) + b
        layer = act(pre) if l + 1 < len(widths) else pre
        sw = q_scale1 if (l == 1 and q_scale1 is not None) else q_scale
        q_vars.append(api.NormalVariable(q_loc_scale * q_scale * rng.normal(0., 1., (rows, 1)), q_scale * np.ones((rows, 1)),
                                         "b%d" % l, learnable=True))
        q_vars.append(api.NormalVariable(q_loc_scale * sw * rng.normal(0., 1., (rows, cols)), sw * np.ones((rows, cols)),
                                         "weights%d" % l, learnable=True))
    # (n_classes = 1: a Bernoulli likelihood on one logit, `BinomialVariable(1, logits=...)` as in
    #  examples/minibatch_logistic_regression.py:27 — the labels are then 0 / 1)
    k = api.CategoricalVariable(logits=layer, name="k") if n_classes > 1 else api.BinomialVariable(1, logits=layer, name="k")
    model = api.ProbabilisticModel([k])
    k.observe(y)
    model.set_posterior_model(api.ProbabilisticModel(q_vars))
    return model


def vae_modules(n_features, latent_size, hidden1, hidden2, seed=0, decoder_sd_head=False):
    """Encoder / decoder networks with the layer plan of `examples/VAE_playground.py:27-62` (there: 784-256-512-(2,2)
    and 2-512-256-784): two ReLU layers, then a mean head and a softplus(+0.1) scale head; the decoder mirrors the
    trunk and ends in the logits.  Plain torch modules with a seeded initialisation, returning dicts keyed like the
    example's ("mean", "sd").  Layers are created in the order l1..l4 / l1..l3 so that a seed fixes every tensor."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    class Encoder(nn.Module):
        def __init__(self, widths):
            super().__init__()
            self.l1 = nn.Linear(widths[0], widths[1])
            self.l2 = nn.Linear(widths[1], widths[2])
            self.l3 = nn.Linear(widths[2], latent_size)      # location head
            self.l4 = nn.Linear(widths[2], latent_size)      # scale head (pre-activation)

        def forward(self, rows):
            hidden = torch.relu(self.l2(torch.relu(self.l1(rows.squeeze(-1)))))
            return {"mean": self.l3(hidden), "sd": F.softplus(self.l4(hidden)) + 0.1}

    class Decoder(nn.Module):
        def __init__(self, widths):
            super().__init__()
            self.l1 = nn.Linear(latent_size, widths[0])
            self.l2 = nn.Linear(widths[0], widths[1])
            self.l3 = nn.Linear(widths[1], widths[2])
            if decoder_sd_head:                              # a heteroscedastic decoder: a second head for the likelihood's scale
                self.l4 = nn.Linear(widths[1], widths[2])

        def forward(self, code):
            hidden = torch.relu(self.l2(torch.relu(self.l1(code))))
            if decoder_sd_head:
                return {"mean": self.l3(hidden), "sd": F.softplus(self.l4(hidden)) + 0.05}
            return {"mean": self.l3(hidden)}

    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    enc, dec = Encoder((n_features, hidden2, hidden1)), Decoder((hidden1, hidden2, n_features))
    torch.random.set_rng_state(state)
    return enc, dec


def vae_data(dataset_size, n_features, seed=0, real=False):
    """synthetic binary images (MNIST is not available offline): rand > 0.5, stored [DS, P, 1] like
    `VAE_playground.py:24-26`; `real`: standard-normal rows for the Normal-likelihood variant"""
    rng = np.random.RandomState(seed)
    if real:
        return rng.randn(dataset_size, n_features, 1).astype("float32")
    return (rng.rand(dataset_size, n_features, 1) > 0.5).astype("int32")


def build_vae(api, dataset_size=64, batch_size=16, n_features=784, latent_size=2, hidden1=512, hidden2=256, seed=0,
              likelihood="binomial", likelihood_scale=0.5, learnable_prior=False, learnable_likelihood_scale=False):
    """BASELINE config 5: `examples/VAE_playground.py:64-79` — amortised Normal posterior over a latent code,
    Binomial(1, logits = decoder(z)) likelihood, every Monte-Carlo sample drawing its own minibatch.
    Variants of the same pattern: `likelihood="normal"` (real-valued rows, Normal(decoder(z), likelihood_scale) — a number
    or one value per feature; `learnable_likelihood_scale=True`: a learnable parameter of the joint model),
    `learnable_prior=True` (the prior's loc and scale are learnable parameters of the joint model,
    `standard_variables.py:57-68`)."""
    BF = api.BF
    dataset = vae_data(dataset_size, n_features, seed, real=(likelihood == "normal"))
    enc, dec = vae_modules(n_features, latent_size, hidden1, hidden2, seed, decoder_sd_head=(likelihood == "normal" and
                                                                                          isinstance(likelihood_scale, str)))
    encoder = BF.BrancherFunction(enc)
    decoder = BF.BrancherFunction(dec)
    z = api.NormalVariable(np.zeros((latent_size,)), np.ones((latent_size,)), name="z", learnable=learnable_prior)
    decoder_output = api.DeterministicVariable(decoder(z), name="decoder_output")
    if likelihood == "normal" and isinstance(likelihood_scale, str):      # "decoder": the scale is the decoder's second head
        x = api.NormalVariable(decoder_output["mean"], decoder_output["sd"], name="x")
    elif likelihood == "normal":
        if not np.isscalar(likelihood_scale):
            likelihood_scale = np.asarray(likelihood_scale, dtype=np.float64)
        # (learnable=True turns the numeric argument — the scale — into a learnable root of the joint model behind softplus)
        x = api.NormalVariable(decoder_output["mean"], likelihood_scale, name="x", learnable=bool(learnable_likelihood_scale))
    else:
        x = api.BinomialVariable(total_count=1, logits=decoder_output["mean"], name="x")
    model = api.ProbabilisticModel([x, z])
    Qx = api.EmpiricalVariable(dataset, batch_size=batch_size, name="x", is_observed=True)
    encoder_output = api.DeterministicVariable(encoder(Qx), name="encoder_output")
    Qz = api.NormalVariable(encoder_output["mean"], encoder_output["sd"], name="z")
    model.set_posterior_model(api.ProbabilisticModel([Qx, Qz]))
    model.vae_modules = (enc, dec)
    return model
