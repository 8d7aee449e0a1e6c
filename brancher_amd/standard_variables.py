"""
Standard random variables (`brancher/standard_variables.py`).

Same constructors and the same auto-parameterisation rule (`standard_variables.py:57-68`):
every constructor argument given as a number/array becomes a ``RootVariable`` named
``"<var>_<arg>"`` that stores the *unconstrained* value
(``range.inverse_transform``), and the link uses ``range.forward_transform(root)``.

``LogitNormalVariable`` is named by the README (`README.md:30,56`) but commented out in the
reference snapshot (`standard_variables.py:201-213`, no ``LogitNormalDistribution``
exists).  It is provided here as the reparameterisation SURVEY §8c defines: a Normal latent
on the logit scale whose *use in links* is ``sigmoid(u)``.
"""
import numbers

import numpy as np

import brancher_amd.distributions as distributions
import brancher_amd.functions as BF
import brancher_amd.geometric_ranges as geometric_ranges
from brancher_amd.variables import var2link, Variable, RootVariable, RandomVariable, PartialLink, _Observation


class LinkConstructor:
    """Named parameter links of one variable (`standard_variables.py:14-29`)."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.modules = [link for partial_link in kwargs.values()
                        for link in getattr(var2link(partial_link), "links", ())]

    def expressions(self):
        return {k: var2link(x) for k, x in self.kwargs.items()}

    def parameters(self):
        out = []
        for m in self.modules:
            out.extend(list(m.parameters()))
        return out

    def __iter__(self):
        return iter(self.modules)


def _leading_dimension(value):
    """what the range transforms take as `dim`: the leading axis of an array, 1 for a number, [] otherwise"""
    if isinstance(value, np.ndarray):
        return value.shape[0]
    return 1 if isinstance(value, numbers.Number) else []


class VariableConstructor(RandomVariable):
    """Base of the standard variables (`standard_variables.py:32-68`).  A distribution parameter given as a number or an
    array becomes a RootVariable named ``<variable>_<parameter>`` holding the UNCONSTRAINED value
    (``range.inverse_transform``), and the parameter's link is ``range.forward_transform`` of that root — so a learnable
    scale is optimised on the softplus-inverse scale, a probability on the logit scale (`geometric_ranges.py`).
    Parameters given as variables or links are used as they are."""

    def __init__(self, name, learnable, ranges, is_observed=False, **kwargs):
        self.name = name
        self._observation = _Observation(declared=is_observed)
        links = {parameter: self._parameter_link(parameter, value, ranges, learnable, is_observed)
                 for parameter, value in kwargs.items()}
        self.partial_links = {parameter: var2link(link) for parameter, link in links.items()}
        self.parents = set()
        for link in self.partial_links.values():
            if isinstance(link, PartialLink):
                self.parents |= link.vars
        self.ancestors = set(self.parents)
        for parent in self.parents:
            self.ancestors |= parent.ancestors
        self.link = LinkConstructor(**links)
        self.ranges = {}
        self.is_normalized = True

    def _parameter_link(self, parameter, value, ranges, learnable, is_observed):
        if isinstance(value, (Variable, PartialLink)):
            return value
        dim = _leading_dimension(value)
        root = RootVariable(ranges[parameter].inverse_transform(value, dim), "{}_{}".format(self.name, parameter),
                            learnable, is_observed=is_observed)
        return ranges[parameter].forward_transform(root, dim)


class DeterministicVariable(VariableConstructor):
    # `standard_variables.py:115-130`

    def __init__(self, value, name, learnable=False, is_observed=False,
                 variable_range=geometric_ranges.UnboundedRange()):
        self._type = "Deterministic node"
        ranges = {"value": variable_range}
        super().__init__(name, value=value, learnable=learnable, ranges=ranges, is_observed=is_observed)
        self.distribution = distributions.DeterministicDistribution()

    @property
    def value(self):
        return self._get_sample(1)[self]


def _loc_scale_ranges():
    return {"loc": geometric_ranges.UnboundedRange(), "scale": geometric_ranges.RightHalfLine(0.)}


class NormalVariable(VariableConstructor):
    # `standard_variables.py:133-145`

    def __init__(self, loc, scale, name, learnable=False, is_observed=False):
        self._type = "Normal"
        super().__init__(name, loc=loc, scale=scale, learnable=learnable, ranges=_loc_scale_ranges(),
                         is_observed=is_observed)
        self.distribution = distributions.NormalDistribution()


class CauchyVariable(VariableConstructor):
    # `standard_variables.py:156-168`

    def __init__(self, loc, scale, name, learnable=False, is_observed=False):
        self._type = "Cauchy"
        super().__init__(name, loc=loc, scale=scale, learnable=learnable, ranges=_loc_scale_ranges(),
                         is_observed=is_observed)
        self.distribution = distributions.CauchyDistribution()


class LaplaceVariable(VariableConstructor):
    # `standard_variables.py:171-183`

    def __init__(self, loc, scale, name, learnable=False, is_observed=False):
        self._type = "Laplace"
        super().__init__(name, loc=loc, scale=scale, learnable=learnable, ranges=_loc_scale_ranges(),
                         is_observed=is_observed)
        self.distribution = distributions.LaplaceDistribution()


class LogNormalVariable(VariableConstructor):
    # `standard_variables.py:186-198`

    def __init__(self, loc, scale, name, learnable=False, is_observed=False):
        self._type = "Log Normal"
        super().__init__(name, loc=loc, scale=scale, learnable=learnable, ranges=_loc_scale_ranges(),
                         is_observed=is_observed)
        self.distribution = distributions.LogNormalDistribution()


class LogitNormalVariable(NormalVariable):
    """README `README.md:30,56`; SURVEY §8c definition.  The variable itself is the Normal
    latent ``u`` on the logit scale; arithmetic on it (``b * x``) uses ``sigmoid(u)``.
    With the semi-analytic ELBO of the reference (`variables.py:851-855`) this equals a
    true logit-normal with ``-log q`` entropy in expectation (the Jacobians cancel)."""

    def __init__(self, loc, scale, name, learnable=False, is_observed=False):
        super().__init__(loc, scale, name, learnable=learnable, is_observed=is_observed)
        self._type = "Logit Normal"

    def _apply_operator(self, other, op):
        return BF.sigmoid(self)._apply_operator(other, op)


class BetaVariable(VariableConstructor):
    # `standard_variables.py:216-231` (the reference labels the type "Logit Normal" by mistake)

    def __init__(self, alpha, beta, name, learnable=False, is_observed=False):
        self._type = "Beta"
        ranges = {"concentration1": geometric_ranges.RightHalfLine(0.),
                  "concentration0": geometric_ranges.RightHalfLine(0.)}
        super().__init__(name, concentration1=alpha, concentration0=beta,
                         learnable=learnable, ranges=ranges, is_observed=is_observed)
        self.distribution = distributions.BetaDistribution()


class BinomialVariable(VariableConstructor):
    # `standard_variables.py:234-255`

    def __init__(self, total_count, probs=None, logits=None, name="Binomial", learnable=False, is_observed=False):
        self._type = "Binomial"
        if probs is not None and logits is None:
            ranges = {"total_count": geometric_ranges.UnboundedRange(),
                      "probs": geometric_ranges.Interval(0., 1.)}
            super().__init__(name, total_count=total_count, probs=probs, learnable=learnable, ranges=ranges,
                             is_observed=is_observed)
        elif logits is not None and probs is None:
            ranges = {"total_count": geometric_ranges.UnboundedRange(),
                      "logits": geometric_ranges.UnboundedRange()}
            super().__init__(name, total_count=total_count, logits=logits, learnable=learnable, ranges=ranges)
        else:
            raise ValueError("Either probs or " + "logits needs to be provided as input")
        self.distribution = distributions.BinomialDistribution()


class BernulliVariable(VariableConstructor):
    # `standard_variables.py:258-277` (spelling as in the reference)

    def __init__(self, probs=None, logits=None, name="Bernulli", learnable=False, is_observed=False):
        self._type = "Bernulli"
        if probs is not None and logits is None:
            ranges = {"probs": geometric_ranges.Interval(0., 1.)}
            super().__init__(name, probs=probs, learnable=learnable, ranges=ranges, is_observed=is_observed)
        elif logits is not None and probs is None:
            ranges = {"logits": geometric_ranges.UnboundedRange()}
            super().__init__(name, logits=logits, learnable=learnable, ranges=ranges)
        else:
            raise ValueError("Either probs or " + "logits needs to be provided as input")
        self.distribution = distributions.BernulliDistribution()


BernoulliVariable = BernulliVariable


class CategoricalVariable(VariableConstructor):
    # `standard_variables.py:280-299`

    def __init__(self, probs=None, logits=None, name="Categorical", learnable=False, is_observed=False):
        self._type = "Categorical"
        if probs is not None and logits is None:
            # the reference keys the range as "p" (`standard_variables.py:290`), so a constant
            # `probs` raises KeyError there; a Variable/link works.  Same here.
            ranges = {"p": geometric_ranges.Simplex()}
            super().__init__(name, probs=probs, learnable=learnable, ranges=ranges, is_observed=is_observed)
        elif logits is not None and probs is None:
            ranges = {"logits": geometric_ranges.UnboundedRange()}
            super().__init__(name, logits=logits, learnable=learnable, ranges=ranges, is_observed=is_observed)
        else:
            raise ValueError("Either probs or " + "logits needs to be provided as input")
        self.distribution = distributions.CategoricalDistribution()


class MultivariateNormalVariable(VariableConstructor):
    # `standard_variables.py:317-347` — graph construction only; MVN kernels are a later row
    # (SURVEY §8f-4).

    def __init__(self, loc, covariance_matrix=None, precision_matrix=None,
                 scale_tril=None, name="Multivariate Normal", learnable=False, is_observed=False):
        self._type = "Multivariate Normal"
        given = [x is not None for x in (scale_tril, covariance_matrix, precision_matrix)]
        if sum(given) != 1:
            raise ValueError("Either covariance_matrix or precision_matrix or" +
                             "scale_tril needs to be provided as input")
        if scale_tril is not None:
            ranges = {"loc": geometric_ranges.UnboundedRange(), "scale_tril": geometric_ranges.UnboundedRange()}
            super().__init__(name, loc=loc, scale_tril=scale_tril, learnable=learnable, ranges=ranges,
                             is_observed=is_observed)
        elif covariance_matrix is not None:
            ranges = {"loc": geometric_ranges.UnboundedRange(),
                      "covariance_matrix": geometric_ranges.PositiveDefiniteMatrix()}
            super().__init__(name, loc=loc, covariance_matrix=covariance_matrix, learnable=learnable,
                             ranges=ranges, is_observed=is_observed)
        else:
            ranges = {"loc": geometric_ranges.UnboundedRange(),
                      "precision_matrix": geometric_ranges.UnboundedRange()}
            super().__init__(name, loc=loc, precision_matrix=precision_matrix, learnable=learnable,
                             ranges=ranges, is_observed=is_observed)
        self.distribution = distributions.MultivariateNormalDistribution()


class EmpiricalVariable(VariableConstructor):
    """A minibatch of rows of a dataset (`standard_variables.py:71-96`).  Which rows: `indices` (a list, or a
    `RandomIndices` variable shared with other empirical variables) or, per Monte-Carlo sample, `batch_size` rows drawn
    without replacement (`distributions.py:436-441`).  On the device the rows are never copied out: the consuming GEMM
    gathers them from the HBM-resident dataset."""

    _optional = ("batch_size", "indices", "weights")

    def __init__(self, dataset, name, learnable=False, is_observed=False, batch_size=None, indices=None,
                 weights=None):
        self._type = "Empirical"
        supplied = dict(dataset=dataset)
        for key, value in zip(self._optional, (batch_size, indices, weights)):
            if value is not None:
                supplied[key] = value
        super().__init__(name, learnable=learnable, is_observed=is_observed,
                         ranges=dict.fromkeys(supplied, geometric_ranges.UnboundedRange()), **supplied)
        self.batch_size = self._rows_per_sample(batch_size, indices)
        self.distribution = distributions.EmpiricalDistribution(batch_size=self.batch_size, is_observed=is_observed)

    @staticmethod
    def _rows_per_sample(batch_size, indices):
        """an explicit batch size wins; otherwise a non-empty index collection fixes it (an index *variable* has a
        `__len__`: `RandomIndices` below)"""
        if batch_size:
            return batch_size
        if indices:
            return len(indices)
        raise ValueError("Either the indices or the batch size has to be given as input")


class RandomIndices(EmpiricalVariable):
    """`batch_size` distinct positions of `range(dataset_size)` per draw (`standard_variables.py:99-112`): the index
    variable several `EmpiricalVariable`s of one model share so that features and labels see the same rows."""

    def __init__(self, dataset_size, batch_size, name, is_observed=False):
        positions = list(range(dataset_size))
        super().__init__(positions, name, is_observed=is_observed, batch_size=batch_size)
        self._type = "Random Index"

    def __len__(self):
        return self.batch_size
