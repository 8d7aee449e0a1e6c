"""
Gradient estimators (`brancher/gradient_estimators.py`).

In the reference an estimator receives an opaque closure and a sampler and is evaluated in
Python (`gradient_estimators.py:17-44`).  Here the classes are *selectors*: both land on the
same fused kernel (include/bsvi.h `bsvi_estimator`), which evaluates

  Pathwise :  mean_s f(z_s),                      z = reparameterised draw      (:39-44)
  BlackBox :  mean_s [ log q(z_s) * stopgrad(f(z_s)) + f(z_s) ]                  (:29-36)
  Taylor1  :  mean_s f(E[q | sampled parents]_s)  — a different program, same accumulation  (:47-56)

BlackBox reproduces the reference exactly, including that its value is not the ELBO and that
reparameterisable nodes still carry the pathwise term (the ``differentiable=False`` flag is
dropped at `variables.py:567`).

A USER-DEFINED subclass (the reference's seam: ctor ``(function, sampler, empirical_samples)``, ``__call__(n_samples)``
returning a scalar) runs as written: ``self.sampler._get_sample(n)`` hands out the draw as an opaque object,
``self.function(samples)`` and ``self.sampler.calculate_log_probability(samples)`` return the per-sample f and log q
as torch tensors [N, 1] on the device, and whatever scalar the estimator forms from them is differentiated through two
passes of the fused kernel (`engine.custom_estimator_loss`): e.g. a score-function estimator with a baseline,

    class Baseline(GradientEstimator):
        def __call__(self, n):
            s = self.sampler._get_sample(n, differentiable=False); s.update(self.empirical_samples)
            f = self.function(s)
            return (self.sampler.calculate_log_probability(s) * (f - f.mean()).detach() + f).mean()
"""
from abc import ABC, abstractmethod


class GradientEstimator(ABC):
    kernel_name = None

    def __init__(self, function=None, sampler=None, empirical_samples={}):
        self.function = function
        self.sampler = sampler
        self.empirical_samples = empirical_samples

    @abstractmethod
    def __call__(self, n_samples):
        pass


class _Fused(GradientEstimator):
    def __call__(self, n_samples):
        raise NotImplementedError("fused estimators are evaluated by the engine: use "
                                  "ProbabilisticModel.estimate_log_model_evidence(..., gradient_estimator=cls)")


class PathwiseDerivativeEstimator(_Fused):
    kernel_name = "pathwise"


class BlackBoxEstimator(_Fused):
    kernel_name = "blackbox"


class Taylor1Estimator(_Fused):
    # `gradient_estimators.py:47-56`: f at the analytic means of the posterior given the sampled parents.  Lowered for
    # Normal posteriors (lowering._Lowering.mean_value); the reference itself raises for Bernoulli latents under
    # torch >= 1.8 (validate_args rejects the value 0.5).
    kernel_name = "taylor1"
