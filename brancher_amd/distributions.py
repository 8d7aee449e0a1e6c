"""
Distribution descriptors.

In the reference every node owns a ``Distribution`` object whose template methods build a
fresh ``torch.distributions.X(**params)`` per call and dispatch ``rsample / log_prob /
entropy`` to it (`brancher/distributions.py:63-181`).  Here the arithmetic lives in the
fused HIP kernel (`csrc/dist_math.h`); these classes only carry what the lowering needs:
the kernel's distribution id, the parameter names in kernel order and the reference's
capability flags (which decide e.g. analytic entropy vs the ``-log q`` fallback,
`brancher/variables.py:156-162`).
"""

# kernel distribution ids — must match enum bsvi_dist in include/bsvi.h
DIST_DETERMINISTIC = 0
DIST_NORMAL = 1
DIST_LOGNORMAL = 2
DIST_CAUCHY = 3
DIST_LAPLACE = 4
DIST_BETA = 5
DIST_BINOMIAL = 6
DIST_BERNOULLI = 7
DIST_CATEGORICAL = 8
DIST_LINEAR = 9            # BSVI_DIST_LINEAR: not a distribution — the surrogate of a term computed outside the program
DIST_EMPIRICAL = 100       # host-side kinds (they never reach an instruction)
DIST_MVNORMAL = 101

# what the per-latent "noise" input means in given-noise (parity) mode
NOISE_NONE = 0          # deterministic
NOISE_STD_NORMAL = 1    # eps ~ N(0,1); z = loc + scale*eps (LogNormal: exp of that)
NOISE_STD_CAUCHY = 2    # eps ~ Cauchy(0,1)
NOISE_UNIFORM_PM1 = 3   # u ~ U(eps-1, 1) (torch laplace.py:83)
NOISE_VALUE = 4         # the draw itself is supplied (Beta, discrete variables)


class Distribution:
    kind = None
    noise = NOISE_NONE

    def __init__(self):
        self.required_parameters = set()
        # reference defaults are None (`distributions.py:35-40`)
        self.has_differentiable_samples = None
        self.is_finite = None
        self.is_discrete = None
        self.has_analytic_entropy = None
        self.has_analytic_mean = None
        self.has_analytic_var = None

    #: order in which the kernel expects the parameters (p0, p1)
    kernel_parameters = ()

    def check_parameters(self, **parameters):
        # `distributions.py:43-45`
        assert all([any([p in parameters for p in tpl]) if isinstance(tpl, tuple) else tpl in parameters
                    for tpl in self.required_parameters])

    def resolve_kernel_parameters(self, parameters):
        """Pick which named link outputs feed kernel operands p0/p1.  Returns a list of
        (name, transform) where transform is None or a string understood by the lowering."""
        return [(name, None) for name in self.kernel_parameters]


class LinearSurrogate(Distribution):
    """"log-density" p0 * x + p1 (include/bsvi.h BSVI_DIST_LINEAR): how the batched multivariate-normal kernel's result
    enters the per-sample program (lowering.mvn_external).  Never built by a user."""
    kind = DIST_LINEAR
    kernel_parameters = ("coefficient", "offset")


class DeterministicDistribution(Distribution):
    kind = DIST_DETERMINISTIC
    kernel_parameters = ("value",)

    def __init__(self):
        super().__init__()
        self.required_parameters = {"value"}
        self.has_differentiable_samples = True
        self.is_finite = True
        self.is_discrete = True
        self.has_analytic_entropy = True
        self.has_analytic_mean = True
        self.has_analytic_var = True


class _LocScale(Distribution):
    kernel_parameters = ("loc", "scale")

    def __init__(self):
        super().__init__()
        self.required_parameters = {"loc", "scale"}
        self.has_differentiable_samples = True
        self.is_finite = False
        self.is_discrete = False
        self.has_analytic_entropy = True
        self.has_analytic_mean = True
        self.has_analytic_var = True


class NormalDistribution(_LocScale):
    kind = DIST_NORMAL
    noise = NOISE_STD_NORMAL


class LogNormalDistribution(_LocScale):
    kind = DIST_LOGNORMAL
    noise = NOISE_STD_NORMAL


class CauchyDistribution(_LocScale):
    kind = DIST_CAUCHY
    noise = NOISE_STD_CAUCHY

    def __init__(self):
        super().__init__()
        self.has_analytic_var = False


class LaplaceDistribution(_LocScale):
    kind = DIST_LAPLACE
    noise = NOISE_UNIFORM_PM1


class BetaDistribution(Distribution):
    kind = DIST_BETA
    noise = NOISE_VALUE
    kernel_parameters = ("concentration1", "concentration0")

    def __init__(self):
        super().__init__()
        self.required_parameters = {"concentration1", "concentration0"}
        self.has_differentiable_samples = True
        self.is_finite = False
        self.is_discrete = False
        self.has_analytic_entropy = True
        self.has_analytic_mean = True
        self.has_analytic_var = True


class _ProbsOrLogits(Distribution):
    """Binomial / Bernoulli accept ``probs`` or ``logits``; the kernel works on logits
    (torch converts probs with a clamp, `torch/distributions/utils.py` probs_to_logits)."""

    def resolve_probs(self, parameters):
        if "logits" in parameters:
            return ("logits", None)
        return ("probs", "probs_to_logits")


class BinomialDistribution(_ProbsOrLogits):
    kind = DIST_BINOMIAL
    noise = NOISE_VALUE

    def __init__(self):
        super().__init__()
        self.required_parameters = {"total_count", ("probs", "logits")}
        self.has_differentiable_samples = False
        self.is_finite = True
        self.is_discrete = True
        self.has_analytic_entropy = False
        self.has_analytic_mean = True
        self.has_analytic_var = True

    def resolve_kernel_parameters(self, parameters):
        return [("total_count", None), self.resolve_probs(parameters)]


class BernulliDistribution(_ProbsOrLogits):
    kind = DIST_BERNOULLI
    noise = NOISE_VALUE

    def __init__(self):
        super().__init__()
        self.required_parameters = {("probs", "logits")}
        self.has_differentiable_samples = False
        self.is_finite = True
        self.is_discrete = True
        self.has_analytic_entropy = True
        self.has_analytic_mean = True
        self.has_analytic_var = True

    def resolve_kernel_parameters(self, parameters):
        return [self.resolve_probs(parameters)]


class CategoricalDistribution(Distribution):
    """The reference constructor sets mis-named attributes (`distributions.py:287-292`),
    so the real capability flags keep the base-class ``None``: samples are drawn with
    ``.sample()`` and the entropy falls back to ``-log q``.  Reproduced on purpose."""
    kind = DIST_CATEGORICAL
    noise = NOISE_VALUE

    def __init__(self):
        super().__init__()
        self.required_parameters = {("probs", "logits")}
        self.vector_parameters = {"probs", "logits"}


class MultivariateNormalDistribution(Distribution):
    kind = DIST_MVNORMAL

    def __init__(self):
        super().__init__()
        self.required_parameters = {"loc", ("covariance_matrix", "precision_matrix", "scale_tril")}
        self.has_differentiable_samples = True
        self.is_finite = False
        self.is_discrete = False
        self.has_analytic_entropy = True
        self.has_analytic_mean = True
        self.has_analytic_var = True


class EmpiricalDistribution(Distribution):
    kind = DIST_EMPIRICAL

    def __init__(self, batch_size, is_observed):
        super().__init__()
        self.required_parameters = {"dataset"}
        self.batch_size = batch_size
        self.is_observed = is_observed
        self.has_differentiable_samples = False
        self.is_finite = True
        self.is_discrete = True
        self.has_analytic_entropy = True
        self.has_analytic_mean = False
        self.has_analytic_var = False
