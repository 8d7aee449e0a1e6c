"""
The reference's examples/VAE_playground.py on the MI355X engine: an MLP variational auto-encoder with an amortised
Normal posterior, Binomial(1, logits) likelihood and a fresh minibatch per Monte-Carlo sample.  MNIST is not available
offline, so the "images" are synthetic 28x28 blobs.  Run on a machine with an MI355X:

    python examples/vae_playground.py
"""
import os
import sys
import time

import numpy as np
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brancher_amd.variables import ProbabilisticModel
from brancher_amd.standard_variables import NormalVariable, EmpiricalVariable, BinomialVariable, DeterministicVariable
from brancher_amd import inference, engine
from brancher_amd.inference import ReverseKL
from brancher_amd.gradient_estimators import PathwiseDerivativeEstimator
import brancher_amd.functions as BF

image_size, latent_size = 28 * 28, 2

# synthetic binary images: a bright disc at a random position
rng = np.random.RandomState(0)
yy, xx = np.mgrid[0:28, 0:28]
centres = rng.uniform(6, 22, size=(5000, 2))
dataset = (((yy[None] - centres[:, 0, None, None]) ** 2 + (xx[None] - centres[:, 1, None, None]) ** 2) < 25.0)
dataset = dataset.reshape(-1, image_size, 1).astype("int32")


class EncoderArchitecture(nn.Module):
    def __init__(self, image_size, latent_size, hidden_size1=512, hidden_size2=256):
        super().__init__()
        self.l1 = nn.Linear(image_size, hidden_size2)
        self.l2 = nn.Linear(hidden_size2, hidden_size1)
        self.f1 = nn.ReLU()
        self.f2 = nn.ReLU()
        self.l3 = nn.Linear(hidden_size1, latent_size)
        self.l4 = nn.Linear(hidden_size1, latent_size)
        self.softplus = nn.Softplus()

    def __call__(self, x):
        h0 = self.f1(self.l1(x.squeeze()))
        h1 = self.f2(self.l2(h0))
        return {"mean": self.l3(h1), "sd": self.softplus(self.l4(h1)) + 0.1}


class DecoderArchitecture(nn.Module):
    def __init__(self, latent_size, image_size, hidden_size1=512, hidden_size2=256):
        super().__init__()
        self.l1 = nn.Linear(latent_size, hidden_size1)
        self.l2 = nn.Linear(hidden_size1, hidden_size2)
        self.f1 = nn.ReLU()
        self.f2 = nn.ReLU()
        self.l3 = nn.Linear(hidden_size2, image_size)

    def __call__(self, x):
        return {"mean": self.l3(self.f2(self.l2(self.f1(self.l1(x)))))}


encoder = BF.BrancherFunction(EncoderArchitecture(image_size, latent_size))
decoder = BF.BrancherFunction(DecoderArchitecture(latent_size, image_size))

# generative model
z = NormalVariable(np.zeros((latent_size,)), np.ones((latent_size,)), name="z")
decoder_output = DeterministicVariable(decoder(z), name="decoder_output")
x = BinomialVariable(total_count=1, logits=decoder_output["mean"], name="x")
model = ProbabilisticModel([x, z])

# amortised variational distribution
Qx = EmpiricalVariable(dataset, batch_size=100, name="x", is_observed=True)
encoder_output = DeterministicVariable(encoder(Qx), name="encoder_output")
Qz = NormalVariable(encoder_output["mean"], encoder_output["sd"], name="z")
model.set_posterior_model(ProbabilisticModel([Qx, Qz]))

t0 = time.time()
inference.perform_inference(model, inference_method=ReverseKL(gradient_estimator=PathwiseDerivativeEstimator),
                            number_iterations=1000, number_samples=8, optimizer="Adam", lr=0.001)
loss = model.diagnostics["loss curve"]
print("1000 iterations of 800 rows in %.2f s; loss %.1f -> %.1f" % (time.time() - t0, loss[:20].mean(), loss[-20:].mean()))

# posterior predictive: decode a grid of latent codes (the image grid of the reference example)
compiled = engine.compile_model(model, None, PathwiseDerivativeEstimator)
grid = np.stack(np.meshgrid(np.linspace(-3, 3, 8), np.linspace(-3, 3, 8)), -1).reshape(-1, 2)
probs = 1.0 / (1.0 + np.exp(-np.clip(compiled.decode(grid).cpu().numpy(), -30, 30)))
print("decoded %d latent codes -> images of mean intensity %.3f (data: %.3f)" % (len(grid), probs.mean(), dataset.mean()))
compiled.sync_modules()          # the trained tensors are back in the torch modules
