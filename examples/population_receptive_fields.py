"""
`examples/PopulationReceptiveFields.py` of the reference on the MI355X engine: the response of a Gaussian receptive field — centre
(mu_x, mu_y) and width v latent — to 15 stimulus images on a 40 x 40 mesh,

    mean_response = BF.sum(BF.sum(receptive_field * experimental_input, dim=1, keepdim=True), dim=2, keepdim=True)

a reduction over 1 600 elements per datapoint and Monte-Carlo sample.  The per-sample program does not unroll that; the library's REDUCE
node (bsvi_reduce_*) evaluates it — one workgroup per sample — and hands value and gradient back to the program.  The stimulus node is
observed BY FLAG only (never given a value): like the reference (variables.py:849, 553-565) the engine draws it from its Normal once per
iteration, on the device.

    python examples/population_receptive_fields.py          (needs an MI355X)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brancher_amd.functions as BF
from brancher_amd import inference
from brancher_amd.standard_variables import LogNormalVariable, NormalVariable
from brancher_amd.variables import ProbabilisticModel, RootVariable

S, N, DATAPOINTS = 6., 40, 15
x_range = np.linspace(-S / 2., S / 2., N)
x_mesh, y_mesh = np.meshgrid(x_range, x_range)

# experimental model
x = RootVariable(x_mesh, name="x")
y = RootVariable(y_mesh, name="y")
w1 = NormalVariable(0., 1., name="w1")
w2 = NormalVariable(0., 1., name="w2")
b = NormalVariable(0., 1., name="b")
experimental_input = NormalVariable(BF.exp(BF.sin(w1 * x + w2 * y + b)), 0.1, name="input", is_observed=True)

# probabilistic model
mu_x = NormalVariable(0., 1., name="mu_x")
mu_y = NormalVariable(0., 1., name="mu_y")
v = LogNormalVariable(0., 0.1, name="v")
nu = LogNormalVariable(-1, 0.01, name="nu")
receptive_field = BF.exp((-(x - mu_x) ** 2 - (y - mu_y) ** 2) / (2. * v ** 2)) / (2. * BF.sqrt(np.pi * v ** 2))
mean_response = BF.sum(BF.sum(receptive_field * experimental_input, dim=1, keepdim=True), dim=2, keepdim=True)
response = NormalVariable(mean_response, nu, name="response")
model = ProbabilisticModel([response, experimental_input])

# data: the generative process at mu = (0.6, -0.4), v = 0.5, nu = 0.1 (the reference samples them from the model itself; a centre
# far out in the prior's tail gives a narrow bump the stochastic gradients do not find from the prior's mode)
rng = np.random.RandomState(0)
w1v, w2v, bv = (rng.normal(0., 1., size=(DATAPOINTS, 1)) for _ in range(3))
stimulus = np.exp(np.sin(w1v[:, :, None] * x_mesh + w2v[:, :, None] * y_mesh + bv[:, :, None])) + 0.1 * rng.normal(size=(DATAPOINTS, N, N))
true_field = np.exp((-(x_mesh - 0.6) ** 2 - (y_mesh + 0.4) ** 2) / (2. * 0.5 ** 2)) / (2. * np.sqrt(np.pi * 0.5 ** 2))
responses = (true_field * stimulus).sum(axis=(1, 2)).reshape(DATAPOINTS, 1, 1) + 0.1 * rng.normal(size=(DATAPOINTS, 1, 1))
w1.observe(w1v.astype(np.float32))
w2.observe(w2v.astype(np.float32))
b.observe(bv.astype(np.float32))
response.observe(responses.astype(np.float32))

# variational model
Qmu_x = NormalVariable(0., 1., name="mu_x", learnable=True)
Qmu_y = NormalVariable(0., 1., name="mu_y", learnable=True)
Qv = LogNormalVariable(0., 0.1, name="v", learnable=True)
Qnu = LogNormalVariable(-1, 0.01, name="nu", learnable=True)
model.set_posterior_model(ProbabilisticModel([Qmu_x, Qmu_y, Qv, Qnu]))

t0 = time.perf_counter()
inference.perform_inference(model, number_iterations=1500, number_samples=50, optimizer="Adam", lr=0.01)
loss = np.asarray(model.diagnostics["loss curve"])
print("1500 iterations at 50 samples in %.2f s; loss %.1f -> %.1f" % (time.perf_counter() - t0, loss[:20].mean(), loss[-20:].mean()))
# (the posterior's own variables: `model.get_posterior_sample` would also draw the posterior PREDICTIVE of `response`, whose mean is the
#  reduction — sampling programs do not carry the reduce node, the ELBO programs do)
post = model.posterior_model.get_sample(2000)
print("posterior means: mu_x %.3f  mu_y %.3f  v %.3f   (data generated at 0.6, -0.4, 0.5)" % (
    float(post["mu_x"].mean()), float(post["mu_y"].mean()), float(post["v"].mean())))
