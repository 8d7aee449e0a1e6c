"""Quick start: the README state-space model of Brancher on the MI355X engine.

Same modelling code as with the reference (only the package name differs); `perform_inference` runs the whole
optimisation loop in one persistent kernel launch.  Run on a machine with an MI355X:

    python examples/quickstart_autoregressive.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from brancher_amd.variables import ProbabilisticModel
from brancher_amd.standard_variables import NormalVariable, LogitNormalVariable, DeterministicVariable
from brancher_amd import inference
import brancher_amd.functions as BF

T = 20
driving_noise, measure_noise = 1., 0.3

# probabilistic model
x0 = NormalVariable(0., driving_noise, 'x0')
y0 = NormalVariable(x0, measure_noise, 'y0')
b = LogitNormalVariable(0.5, 1., 'b')
x, y = [x0], [y0]
for t in range(1, T):
    x.append(NormalVariable(b * x[t - 1], driving_noise, 'x{}'.format(t)))
    y.append(NormalVariable(x[t], measure_noise, 'y{}'.format(t)))
AR_model = ProbabilisticModel(x + y)

# synthetic observations of y
rng = np.random.RandomState(0)
series = np.zeros(T)
for t in range(1, T):
    series[t] = 0.8 * series[t - 1] + rng.normal(0., driving_noise)
for t, yt in enumerate(y):
    yt.observe(np.array([series[t] + rng.normal(0., measure_noise)], dtype=np.float32))

# structured variational posterior
Qb = LogitNormalVariable(0.5, 0.5, 'b', learnable=True)
logit_b_post = DeterministicVariable(0., 'logit_b_post', learnable=True)
Qx = [NormalVariable(0., 1., 'x0', learnable=True)]
Qx_mean = [DeterministicVariable(0., 'x0_mean', learnable=True)]
for t in range(1, T):
    Qx_mean.append(DeterministicVariable(0., 'x{}_mean'.format(t), learnable=True))
    Qx.append(NormalVariable(BF.sigmoid(logit_b_post) * Qx[t - 1] + Qx_mean[t], 1., 'x{}'.format(t), learnable=True))
AR_model.set_posterior_model(ProbabilisticModel([Qb] + Qx))

t0 = time.time()
inference.perform_inference(AR_model, number_iterations=2000, number_samples=300, optimizer='SGD', lr=0.001)
loss = AR_model.diagnostics["loss curve"]
print("2000 iterations in %.2f s (first call includes lowering and program upload); loss %.2f -> %.2f"
      % (time.time() - t0, loss[0], loss[-1]))

samples = AR_model.get_posterior_sample(2000)
b_post = 1. / (1. + np.exp(-samples["b"].values.astype(np.float64)))       # b lives on the logit scale
print("posterior of the AR coefficient: %.3f +- %.3f (data generated with 0.8)" % (b_post.mean(), b_post.std()))
