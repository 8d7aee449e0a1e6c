"""
An amortised variational auto-encoder on the MI355X engine — the kind of model `examples/VAE_playground.py` of the
reference builds: an encoder network gives the Normal posterior of a 2-d code for every data row, a decoder network
gives the Binomial(1, logits) likelihood, and every Monte-Carlo sample draws its own minibatch.  MNIST is not
available offline; the data here are synthetic 28x28 images of a disc at a random position.

    python examples/amortized_vae.py          (needs an MI355X)
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brancher_amd.functions as BF
from brancher_amd import engine, inference
from brancher_amd.gradient_estimators import PathwiseDerivativeEstimator
from brancher_amd.standard_variables import BinomialVariable, DeterministicVariable, EmpiricalVariable, NormalVariable
from brancher_amd.variables import ProbabilisticModel

SIDE, CODE = 28, 2
PIXELS = SIDE * SIDE


def disc_images(count, seed=0):
    rng = np.random.RandomState(seed)
    rows, cols = np.mgrid[0:SIDE, 0:SIDE]
    cy, cx = rng.uniform(6, 22, size=(2, count, 1, 1))
    inside = (rows[None] - cy) ** 2 + (cols[None] - cx) ** 2 < 25.0
    return inside.reshape(count, PIXELS, 1).astype("int32")


class Recognition(nn.Module):
    """pixels -> parameters of q(code | pixels)"""

    def __init__(self):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(PIXELS, 256), nn.ReLU(), nn.Linear(256, 512), nn.ReLU())
        self.loc, self.spread = nn.Linear(512, CODE), nn.Linear(512, CODE)

    def forward(self, pixels):
        h = self.trunk(pixels.squeeze(-1))
        return {"mean": self.loc(h), "sd": F.softplus(self.spread(h)) + 0.1}


class Generator(nn.Module):
    """code -> logits of every pixel"""

    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(CODE, 512), nn.ReLU(), nn.Linear(512, 256), nn.ReLU(), nn.Linear(256, PIXELS))

    def forward(self, code):
        return {"logits": self.net(code)}


images = disc_images(5000)
recognise, generate = BF.BrancherFunction(Recognition()), BF.BrancherFunction(Generator())

code = NormalVariable(np.zeros((CODE,)), np.ones((CODE,)), name="code")
generated = DeterministicVariable(generate(code), name="generated")
pixels = BinomialVariable(total_count=1, logits=generated["logits"], name="pixels")
model = ProbabilisticModel([pixels, code])

batch = EmpiricalVariable(images, batch_size=100, name="pixels", is_observed=True)
recognised = DeterministicVariable(recognise(batch), name="recognised")
model.set_posterior_model(ProbabilisticModel([batch, NormalVariable(recognised["mean"], recognised["sd"], name="code")]))

start = time.time()
inference.perform_inference(model, inference_method=inference.ReverseKL(gradient_estimator=PathwiseDerivativeEstimator),
                            number_iterations=1000, number_samples=8, optimizer="Adam", lr=0.001)
curve = model.diagnostics["loss curve"]
print("1000 iterations of 800 rows in %.2f s; loss %.1f -> %.1f" % (time.time() - start, curve[:20].mean(), curve[-20:].mean()))

# posterior predictive: decode a grid of codes with the trained generator
compiled = engine.compile_model(model, None, PathwiseDerivativeEstimator)
grid = np.stack(np.meshgrid(np.linspace(-3, 3, 8), np.linspace(-3, 3, 8)), -1).reshape(-1, CODE)
intensity = torch.sigmoid(compiled.decode(grid)).mean().item()
print("decoded %d codes -> mean pixel intensity %.3f (data: %.3f)" % (len(grid), intensity, images.mean()))
compiled.sync_modules()          # the trained tensors are back in the torch modules
