"""Gaussian-process regression with inferred kernel hyper-parameters on the MI355X engine.

f ~ MultivariateNormal(0, K) with the squared-exponential covariance K = amplitude * exp(-sqdist / (2 ell^2)) + jitter I of
eight fixed inputs; the length-scale `ell` is a LogNormal latent, the amplitude exp(a learnable parameter of the joint model)
(type-II maximum likelihood), y ~ Normal(f, 0.2) observed.  The covariance is an ordinary link expression: the engine unrolls
its Cholesky factorisation into the per-sample program.  Run on a machine with an MI355X:

    python examples/gp_hyperparameters.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from brancher_amd.variables import ProbabilisticModel, RootVariable
from brancher_amd.standard_variables import NormalVariable, LogNormalVariable, MultivariateNormalVariable
from brancher_amd import inference
import brancher_amd.functions as BF

n, noise, jitter = 8, 0.2, 3e-2
rng = np.random.RandomState(1)
x = np.linspace(-2., 2., n)
targets = np.sin(2 * np.pi * 0.15 * x) + noise * rng.normal(0., 1., n)

# probabilistic model
sqdist = RootVariable(((x[:, None] - x[None, :]) ** 2).astype(np.float32), "sqdist")
eye = RootVariable((jitter * np.eye(n)).astype(np.float32), "jitter")
ell = LogNormalVariable(-0.5, 0.5, "ell")
log_amplitude = RootVariable(0.0, "log_amplitude", learnable=True)      # (positive through exp: K must stay positive definite)
K = BF.exp(sqdist * (-0.5) / (ell * ell)) * BF.exp(log_amplitude) + eye
f = MultivariateNormalVariable(loc=np.zeros((n,)), covariance_matrix=K, name="f")
y = NormalVariable(f, noise, name="y")
model = ProbabilisticModel([y])
y.observe(targets[None, :].astype(np.float32))

# variational posterior: LogNormal over the length-scale, mean-field Normal over the function values.  Its parameters are
# explicitly named roots: `NormalVariable(..., name="f", learnable=True)` would call them f_loc / f_scale, the names of the
# PRIOR's own parameters, and Brancher maps posterior values onto model variables by name (`utilities.py:282-309`) — the
# prior's mean and the length-scale's prior would silently follow the posterior's
qell_loc = RootVariable(-0.5, "qell_loc", learnable=True)
qell_log_sd = RootVariable(float(np.log(0.3)), "qell_log_sd", learnable=True)
Qell = LogNormalVariable(qell_loc, BF.exp(qell_log_sd), "ell")
qf_mean = RootVariable(np.zeros((n,)), "qf_mean", learnable=True)
qf_log_sd = RootVariable(float(np.log(0.5)), "qf_log_sd", learnable=True)
Qf = NormalVariable(qf_mean, BF.exp(qf_log_sd), name="f")
model.set_posterior_model(ProbabilisticModel([Qell, Qf]))

t0 = time.time()
inference.perform_inference(model, number_iterations=1500, number_samples=256, optimizer="Adam", lr=0.02)
loss = model.diagnostics["loss curve"]
print("1500 SVI iterations at number_samples=256 in %.2f s (lowering and kernel generation included)" % (time.time() - t0))
loss = np.asarray(loss, dtype=np.float64)
print("loss %.2f -> %.2f (%d of %d iterations skipped as non-finite — a draw of a large length-scale can make the 8 x 8\n"
      "      covariance numerically singular in single precision; like `inference.py:98-107` such a step is skipped)"
      % (loss[0], np.nanmean(loss[-50:]), int(np.isnan(loss).sum()), len(loss)))
post = model.get_posterior_sample(2000)
print("posterior length-scale: mean %.3f, sd %.3f" % (post["ell"].mean(), post["ell"].std()))
fm = np.stack([np.asarray(v).reshape(-1) for v in post["f"]]).mean(0)
print("posterior mean of f :", np.round(fm, 2))
print("observations        :", np.round(targets, 2))
if os.environ.get("GP_DEBUG"):
    print("non-finite iterations per block of 100:", [int(np.isnan(loss[i:i + 100]).sum()) for i in range(0, len(loss), 100)])
    print("amplitude:", np.exp(log_amplitude.parameter.numpy().reshape(-1)))
