"""
Inference loop (`brancher/inference.py`).

``perform_inference`` keeps the reference signature and semantics (`inference.py:50-111`):
optimizers for the posterior and — because ``ReverseKL.learnable_model`` is True — for the
joint model; per iteration ``compute_loss`` -> finite check -> ``backward`` -> step of the
first optimizer, the others only when ``iteration > pretraining_iterations``; non-finite
losses warn and skip the step; the loss curve lands in ``diagnostics["loss curve"]``.

What changes is where it runs: the whole iteration is one or two kernel launches
(`engine.CompiledELBO.train`), the finite check and the optimizer step happen on the device
and nothing synchronises with the host inside the loop.  Deviations, on purpose:
  * one loss entry per iteration (the reference appends twice per good iteration,
    `inference.py:105,108`, which makes ``np.array(loss_list)`` raise under numpy >= 1.24);
  * "Numerical error, skipping sample" warnings are emitted after the loop (the flags are
    read back once), not inside it.
"""
import warnings
from abc import ABC, abstractmethod

import numpy as np

from brancher_amd import gradient_estimators
from brancher_amd import engine
from brancher_amd.optimizers import ProbabilisticOptimizer


def perform_inference(joint_model, number_iterations, number_samples=1,
                      optimizer='Adam', input_values={},
                      inference_method=None,
                      posterior_model=None, sampler_model=None,
                      pretraining_iterations=0,
                      **opt_params):
    if not inference_method:
        warnings.warn("The inference method was not specified, using the default reverse KL variational inference")
        inference_method = ReverseKL()
    if not posterior_model:
        posterior_model = joint_model.posterior_model
    if not sampler_model:
        sampler_model = getattr(inference_method, "sampler_model", None) or getattr(joint_model, "posterior_sampler", None)

    joint_model.update_observed_submodel()

    optimizers_list = []
    for model, wanted in ((posterior_model, True), (joint_model, inference_method.learnable_model),
                          (sampler_model, inference_method.learnable_sampler)):
        if wanted and model is not None:
            prob_opt = ProbabilisticOptimizer(model, optimizer, **opt_params)
            if prob_opt.optimizer:
                optimizers_list.append(prob_opt)

    inference_method.check_model_compatibility(joint_model, posterior_model, sampler_model)
    if hasattr(inference_method, "run") and not engine.is_custom_estimator(getattr(inference_method, "gradient_estimator", None)):
        # the methods of this module: the whole loop of `inference.py:95-108` runs on the device
        loss_curve, finite = inference_method.run(joint_model, posterior_model, sampler_model, number_iterations,
                                                  number_samples, optimizer, pretraining_iterations, **opt_params)
        losses = loss_curve.detach().cpu().numpy()
        flags = finite.detach().cpu().numpy()
    else:
        # a reference-style InferenceMethod (only compute_loss / correct_gradient, `inference.py:114-126`): the loop
        # stays here and every step goes through the same device kernels — compute_loss evaluates loss AND gradients in
        # one launch (engine.FusedLoss), ProbabilisticOptimizer.update() is the device optimizer step
        losses, flags = [], []
        for iteration in range(number_iterations):
            loss = inference_method.compute_loss(joint_model, posterior_model, sampler_model, number_samples)
            value = float(loss.detach().cpu())
            flags.append(1.0 if np.isfinite(value) else 0.0)
            if flags[-1]:
                for opt in optimizers_list:
                    opt.zero_grad()
                loss.backward()
                if hasattr(inference_method, "correct_gradient"):
                    inference_method.correct_gradient(joint_model, posterior_model, sampler_model, number_samples)
                for index, opt in enumerate(optimizers_list):
                    if index == 0 or iteration > pretraining_iterations:
                        opt.update()
            losses.append(value)
        losses, flags = np.array(losses, dtype=np.float32), np.array(flags)
    for _ in range(int((flags == 0).sum())):
        warnings.warn("Numerical error, skipping sample")
    joint_model.diagnostics.update({"loss curve": np.array(losses)})
    inference_method.post_process(joint_model)
    # torch modules used as links are trained IN PLACE by the reference (their nn.Parameters belong to the optimizers,
    # `optimizers.py:36-49`); here their tensors live in the engine's parameter buffer: written back
    for link in engine.module_links_of(joint_model, posterior_model):
        link.sync_to_module()


class InferenceMethod(ABC):
    # `inference.py:114-126`

    @abstractmethod
    def check_model_compatibility(self, joint_model, posterior_model, sampler_model):
        pass

    @abstractmethod
    def compute_loss(self, joint_model, posterior_model, sampler_model, number_samples, input_values):
        pass

    @abstractmethod
    def post_process(self, joint_model):
        pass


class ReverseKL(InferenceMethod):
    # `inference.py:129-151`

    def __init__(self, gradient_estimator=gradient_estimators.PathwiseDerivativeEstimator):
        self.learnable_model = True
        self.needs_sampler = False
        self.learnable_sampler = False
        self.gradient_estimator = gradient_estimator

    def check_model_compatibility(self, joint_model, posterior_model, sampler_model):
        pass

    def compute_loss(self, joint_model, posterior_model, sampler_model, number_samples, input_values={}):
        return -joint_model.estimate_log_model_evidence(number_samples=number_samples,
                                                        method="ELBO", input_values=input_values,
                                                        for_gradient=True, posterior_model=posterior_model,
                                                        gradient_estimator=self.gradient_estimator)

    def correct_gradient(self, joint_model, posterior_model, sampler_model, number_samples, input_values={}):
        pass

    def post_process(self, joint_model):
        pass

    def run(self, joint_model, posterior_model, sampler_model, number_iterations, number_samples, optimizer,
            pretraining_iterations, **opt_params):
        """the loop of `inference.py:95-108`, executed by the engine"""
        compiled = engine.compile_model(joint_model, posterior_model, self.gradient_estimator)
        extra = {}
        if getattr(compiled, "prefers_stepwise", None) and compiled.prefers_stepwise(number_samples):
            extra["allow_persistent"] = False
        return compiled.train(number_iterations, number_samples, optimizer,
                              pretraining_iterations=pretraining_iterations, **extra, **opt_params)


class MAP(ReverseKL):
    """Maximum a posteriori point estimates, `inference.py:251-275`: the "posterior" is a model of learnable
    RootVariables carrying the latents' names; the loss is -log p(theta, data).  On this engine that is the ELBO
    program of a posterior without random variables — nothing to sample, no entropy — evaluated on one sample
    (the reference also ignores number_samples here)."""

    def __init__(self):
        super().__init__(gradient_estimator=gradient_estimators.PathwiseDerivativeEstimator)
        self.learnable_model = False
        self.needs_sampler = False
        self.learnable_sampler = False

    def check_model_compatibility(self, joint_model, posterior_model, sampler_model):
        from brancher_amd.variables import RootVariable
        assert all([isinstance(var, RootVariable) for var in posterior_model.flatten()])

    def compute_loss(self, joint_model, posterior_model, sampler_model, number_samples, input_values={}):
        return super().compute_loss(joint_model, posterior_model, sampler_model, 1, input_values)

    def run(self, joint_model, posterior_model, sampler_model, number_iterations, number_samples, optimizer,
            pretraining_iterations, **opt_params):
        return super().run(joint_model, posterior_model, sampler_model, number_iterations, 1, optimizer,
                           pretraining_iterations, **opt_params)
