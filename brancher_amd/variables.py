"""
Graph objects: variables, symbolic links, probabilistic models.

Host-side mirror of `brancher/variables.py` — same class names, constructor signatures,
operator overloading and model bookkeeping — with two structural changes:

* a ``PartialLink`` carries an explicit expression DAG (``symbolic.Expr``) instead of an
  opaque closure (`variables.py:977-1002`), so a (joint, posterior) pair can be lowered
  *once* into a kernel program (`lowering.py`);
* graph objects hold no per-call state.  The reference memoises samples and
  visit-once flags on the variables themselves (``self.samples``, ``self._evaluated``,
  `variables.py:407-409,548-549,504-507`) and clears them with ``reset()``; here the
  compiled program is immutable and all per-call state lives in the engine workspace.

Numerical evaluation never happens in this module: ``_get_sample``,
``calculate_log_probability`` and ``estimate_log_model_evidence`` hand over to the
native engine (`engine.py`), which raises if the HIP library or the GPU is missing.
"""
from abc import ABC, abstractmethod
from collections.abc import Iterable, Hashable
import numbers
import warnings

import numpy as np

from brancher_amd import distributions
from brancher_amd import symbolic as sym
from brancher_amd.modules import Parameter, ParameterModule
from brancher_amd.utilities import coerce_to_dtype, is_discrete, join_sets_list, flatten_list, to_numpy


class BrancherClass(ABC):
    """Abstract superclass of variables, links and models (`variables.py:47-100`)."""

    @abstractmethod
    def _flatten(self):
        pass

    def flatten(self):
        return set(self._flatten())

    def get_variable(self, var_name):
        # `variables.py:68-83`: name -> variable through a dict, so on duplicate names the
        # last one in name-sorted order silently wins.  Kept (the README model itself has
        # a duplicate); the lowering warns about duplicates once per compile.
        flat_list = self._flatten()
        table = {var.name: var for var in flat_list}
        return table[var_name]


class Variable(BrancherClass):
    """Abstract superclass of deterministic and random variables (`variables.py:103-295`)."""

    name = None

    @property
    @abstractmethod
    def is_observed(self):
        pass

    def __str__(self):
        return self.name

    def __repr__(self):
        return "{}({!r})".format(type(self).__name__, self.name)

    # identity semantics: variables are dict keys everywhere, as in the reference
    __hash__ = object.__hash__

    def __eq__(self, other):
        return self is other

    def _apply_operator(self, other, op):
        return var2link(self)._apply_operator(other, op)

    def __neg__(self):
        return -1 * self

    def __add__(self, other):
        return self._apply_operator(other, "add")

    def __radd__(self, other):
        return self.__add__(other)

    def __sub__(self, other):
        return self._apply_operator(other, "sub")

    def __rsub__(self, other):
        return -1 * self.__sub__(other)

    def __mul__(self, other):
        return self._apply_operator(other, "mul")

    def __rmul__(self, other):
        return self.__mul__(other)

    def __truediv__(self, other):
        return self._apply_operator(other, "truediv")

    def __rtruediv__(self, other):
        return self.__truediv__(other) ** (-1)

    def __pow__(self, other):
        return self._apply_operator(other, "pow")

    def __rpow__(self, other):
        raise NotImplementedError

    def __getitem__(self, key):
        # `variables.py:279-289`
        if isinstance(key, str):
            variable_slice = key
        elif isinstance(key, Iterable):
            variable_slice = (slice(None, None, None), *key)
        else:
            variable_slice = (slice(None, None, None), key)
        return PartialLink(vars={self}, expr=sym.Expr("getitem", (sym.variable(self),), variable_slice),
                           links=set(), string="{}[{}]".format(self.name, key))

    def shape(self):
        return PartialLink(vars={self}, expr=sym.Expr("shape", (sym.variable(self),)), links=set())

    # ---- user-facing sampling / statistics: served by the native engine -------------
    def get_sample(self, number_samples, input_values={}):
        from brancher_amd import engine
        return engine.get_sample_frame(self, number_samples, input_values)

    def reset(self, recursive=False):
        pass


class RootVariable(Variable):
    """Constants and learnable parameters (`variables.py:298-381`)."""

    def __init__(self, data, name, learnable=False, is_observed=False):
        self.name = name
        self.distribution = distributions.DeterministicDistribution()
        self._observed = is_observed
        self.parents = set()
        self.ancestors = set()
        self._type = "Deterministic"
        self.learnable = learnable
        self.link = None
        self._value = coerce_to_dtype(data, is_observed)
        if self.learnable:
            if not is_discrete(data):
                self.link = ParameterModule(Parameter(self._value, name=name))
            else:
                self.learnable = False
                warnings.warn('Currently discrete parameters are not learnable. Learnable set to False')

    @property
    def value(self):
        if self.learnable:
            return self.link().numpy()
        return self._value

    @property
    def parameter(self):
        return self.link() if self.learnable else None

    @property
    def is_observed(self):
        return self._observed

    def _flatten(self):
        return []

    def _get_sample(self, number_samples, resample=False, observed=False, input_values={}, differentiable=True):
        from brancher_amd import engine
        return engine.sample_variables([self], number_samples, observed=observed, input_values=input_values)


class RandomVariable(Variable):
    """A node with a distribution, parents and a link (`variables.py:384-622`)."""

    def __init__(self, distribution, name, parents, link):
        self.name = name
        self.distribution = distribution
        self.link = link
        self.parents = parents
        self.ancestors = None
        self._type = "Random"
        self._observed = False
        self._observed_value = None
        self.dataset = None
        self.has_random_dataset = False
        self.has_observed_value = False

    @property
    def value(self):
        if self._observed:
            return self._observed_value
        raise AttributeError('RandomVariable has to be observed to receive value.')

    @property
    def is_observed(self):
        return self._observed

    def observe(self, data):
        # `variables.py:572-590`
        try:
            import pandas as pd
            if isinstance(data, pd.DataFrame):
                from brancher_amd.pandas_interface import pandas_frame2value
                data = pandas_frame2value(data, self.name)
        except ImportError:  # pragma: no cover
            pass
        if isinstance(data, RandomVariable):
            self.dataset = data
            self.has_random_dataset = True
        else:
            self._observed_value = coerce_to_dtype(data, is_observed=True)
            self.has_observed_value = True
        self._observed = True

    def unobserve(self):
        self._observed = False
        self.has_observed_value = False
        self.has_random_dataset = False
        self._observed_value = None
        self.dataset = None

    def _flatten(self):
        variables = list(self.ancestors) + [self]
        return sorted(variables, key=lambda v: v.name)

    def _get_sample(self, number_samples=1, resample=True, observed=False, input_values={}, differentiable=True):
        from brancher_amd import engine
        return engine.sample_variables([self], number_samples, observed=observed, input_values=input_values)

    def calculate_log_probability(self, input_values, reevaluate=True, for_gradient=False,
                                  include_parents=True, normalized=True):
        from brancher_amd import engine
        return engine.log_probability([self], input_values, include_parents=include_parents)


class ProbabilisticModel(BrancherClass):
    """A collection of variables (`variables.py:625-881`)."""

    def __init__(self, variables):
        self._input_variables = self._validate_variables(variables)
        self.variables = self.flatten()
        self.posterior_model = None
        self.posterior_sampler = None
        self.observed_submodel = None
        self.is_transformed = False
        self.diagnostics = {}
        self._compiled = {}     # (posterior id, estimator, ...) -> engine.CompiledELBO
        if not all([var.is_observed for var in self._input_variables]):
            self.update_observed_submodel()
        else:
            self.observed_submodel = self

    @staticmethod
    def _validate_variables(variables):
        for var in variables:
            if not isinstance(var, (RootVariable, RandomVariable, ProbabilisticModel)):
                raise ValueError("Invalid input type: {}".format(type(var)))
        return variables

    def __str__(self):
        return str(self.model_summary)

    @property
    def model_summary(self):
        from brancher_amd.pandas_interface import reformat_model_summary
        var_list = self.flatten()
        return reformat_model_summary([[v._type, v.parents, v.is_observed] for v in var_list],
                                      [v.name for v in var_list], ["Distribution", "Parents", "Observed"])

    @property
    def is_observed(self):
        return all([var.is_observed for var in self._flatten()])

    def _flatten(self):
        variables = list(join_sets_list([var.ancestors.union({var}) for var in self._input_variables]))
        return sorted(variables, key=lambda v: v.name)

    def observe(self, data):
        # `variables.py:681-693`
        try:
            import pandas as pd
            if isinstance(data, pd.DataFrame):
                from brancher_amd.pandas_interface import pandas_frame2value
                data = {var_name: pandas_frame2value(data, index=var_name) for var_name in data}
        except ImportError:  # pragma: no cover
            pass
        if isinstance(data, dict):
            if all([isinstance(k, Variable) for k in data.keys()]):
                data_dict = data
            elif all([isinstance(k, str) for k in data.keys()]):
                data_dict = {self.get_variable(name): value for name, value in data.items()}
            else:
                raise ValueError("The keys of the data dictionary should be all variables or all names")
        else:
            raise ValueError("The input data should be either a dictionary of values or a pandas dataframe")
        for var in data_dict:
            if isinstance(var, RandomVariable):
                var.observe(data_dict[var])

    def update_observed_submodel(self):
        flattened_model = self._flatten()
        observed_variables = [var for var in flattened_model if var.is_observed]
        self.observed_submodel = ProbabilisticModel(observed_variables)

    def set_posterior_model(self, model, sampler=None):
        # `variables.py:703-716`
        self.posterior_model = PosteriorModel(posterior_model=model, joint_model=self)
        self._compiled = {}
        if sampler:
            if isinstance(sampler, ProbabilisticModel):
                self.posterior_sampler = PosteriorModel(sampler, joint_model=self)
            elif isinstance(sampler, Variable):
                self.posterior_sampler = PosteriorModel(ProbabilisticModel([sampler]), joint_model=self)
            elif isinstance(sampler, Iterable) and all([isinstance(s, (ProbabilisticModel, Variable))
                                                        for s in sampler]):
                self.posterior_sampler = [PosteriorModel(ProbabilisticModel([var]), joint_model=self)
                                          if isinstance(var, Variable) else PosteriorModel(var, joint_model=self)
                                          for var in sampler]
            else:
                raise ValueError("The sampler should be ither a probabilistic model, a brancher variable "
                                 "or an iterable of variables and/or models")

    def check_posterior_model(self):
        if not self.posterior_model:
            raise AttributeError("The posterior model has not been initialized.")

    def reset(self):
        pass

    # ---- numerical entry points: all served by the native engine --------------------
    def _get_sample(self, number_samples, observed=False, input_values={}, differentiable=True):
        from brancher_amd import engine
        return engine.sample_model(self, number_samples, observed=observed, input_values=input_values)

    def get_sample(self, number_samples, input_values={}):
        from brancher_amd import engine
        return engine.get_sample_frame(self, number_samples, input_values)

    def _get_posterior_sample(self, number_samples, input_values={}, differentiable=True):
        self.check_posterior_model()
        from brancher_amd import engine
        return engine.posterior_sample(self, number_samples, input_values=input_values)

    def get_posterior_sample(self, number_samples, input_values={}):
        self.check_posterior_model()
        from brancher_amd import engine
        return engine.get_posterior_sample_frame(self, number_samples, input_values)

    def calculate_log_probability(self, rv_values, for_gradient=False, normalized=True):
        from brancher_amd import engine
        return engine.log_probability(self._input_variables, rv_values, include_parents=True, model=self)

    def get_importance_weights(self, q_samples, q_model, empirical_samples={}, for_gradient=False,
                               give_normalization=False):
        """Self-normalised importance weights of posterior samples, `variables.py:821-841`:
        w_n ∝ exp(log p(z_n, y) − log q(z_n)).  The two log-densities come from the fused kernel
        (`engine.importance_log_weights`); the max-shifted softmax over samples is host arithmetic exactly as in
        the reference.  Returns a numpy array [N, 1] (and log normalisation when asked)."""
        import numpy as np
        from brancher_amd import engine
        log_p, log_q = engine.importance_log_weights(self, q_model, q_samples)
        log_weights = (log_p - log_q).detach().cpu().numpy().reshape(-1, 1)
        alpha = np.max(log_weights)
        weights = np.exp(log_weights - alpha)
        norm = np.sum(weights)
        weights /= norm
        if not give_normalization:
            return weights
        return weights, np.log(norm) + alpha

    def estimate_log_model_evidence(self, number_samples, method="ELBO", input_values={},
                                    for_gradient=False, posterior_model=(), gradient_estimator=None):
        # `variables.py:843-870`
        if not posterior_model:
            self.check_posterior_model()
            posterior_model = self.posterior_model
        if method != "ELBO":
            raise NotImplementedError("The requested estimation method is currently not implemented.")
        from brancher_amd import engine
        return engine.estimate_elbo(self, posterior_model, number_samples,
                                    for_gradient=for_gradient, gradient_estimator=gradient_estimator)


class PosteriorModel(ProbabilisticModel):
    """`variables.py:884-907`.  ``_input_variables`` is the *set* of all flattened variables
    of the given model (roots included), exactly as the reference passes
    ``posterior_model.variables`` (`variables.py:894`)."""

    def __init__(self, posterior_model, joint_model):
        super().__init__(posterior_model.variables)
        self.posterior_model = None
        self.joint_model = joint_model
        self._is_trained = False

    @property
    def model_mapping(self):
        from brancher_amd.lowering import get_model_mapping
        return get_model_mapping(self, self.joint_model)


def var2link(var):
    # `variables.py:910-922`
    if isinstance(var, Variable):
        return PartialLink(vars={var}, expr=sym.variable(var), links=set(), string=str(var))
    if sym.is_numeric_constant(var):
        return PartialLink(vars=set(), expr=sym.const(var), links=set(), string=str(var))
    try:
        import torch
        if torch.is_tensor(var):
            return PartialLink(vars=set(), expr=sym.const(to_numpy(var)), links=set(), string="tensor")
    except ImportError:  # pragma: no cover
        pass
    if isinstance(var, (tuple, list)) and all([isinstance(v, (Variable, PartialLink)) for v in var]):
        links = [var2link(v) for v in var]
        return PartialLink(vars=join_sets_list([l.vars for l in links]),
                           expr=sym.Expr("tuple", tuple(l.expr for l in links)),
                           links=join_sets_list([l.links for l in links]), string=str(var))
    return var


class PartialLink(BrancherClass):
    """A symbolic operation between variables (`variables.py:977-1072`)."""

    def __init__(self, vars, expr, links, string=""):
        self.vars = vars
        self.expr = expr
        self.links = links
        self.string = string

    def __str__(self):
        return self.string

    def _apply_operator(self, other, op):
        other = var2link(other)
        if not isinstance(other, PartialLink):
            raise TypeError("Unsupported operand for a symbolic link: {!r}".format(other))
        return PartialLink(vars=self.vars.union(other.vars),
                           expr=sym.binary(op, self.expr, other.expr),
                           links=self.links.union(other.links),
                           string="(" + str(self) + sym.BINARY_SYMBOLS[op] + str(other) + ")")

    def __neg__(self):
        return -1 * self

    def __add__(self, other):
        return self._apply_operator(other, "add")

    def __radd__(self, other):
        return self.__add__(other)

    def __sub__(self, other):
        return self._apply_operator(other, "sub")

    def __rsub__(self, other):
        return -1 * self.__sub__(other)

    def __mul__(self, other):
        return self._apply_operator(other, "mul")

    def __rmul__(self, other):
        return self.__mul__(other)

    def __truediv__(self, other):
        return self._apply_operator(other, "truediv")

    def __rtruediv__(self, other):
        return self.__truediv__(other) ** (-1)

    def __pow__(self, other):
        return self._apply_operator(other, "pow")

    def __rpow__(self, other):
        raise NotImplementedError

    def __getitem__(self, key):
        # `variables.py:1037-1053`
        if isinstance(key, Iterable) and not isinstance(key, str) and all([isinstance(k, int) for k in key]):
            variable_slice = (slice(None, None, None), *key)
        elif isinstance(key, int):
            variable_slice = (slice(None, None, None), key)
        elif isinstance(key, Hashable):
            variable_slice = key
        else:
            raise ValueError("The input to __getitem__ is neither numeric nor a hashabble key")
        return PartialLink(vars=self.vars, expr=sym.Expr("getitem", (self.expr,), variable_slice),
                           links=self.links, string="{}[{}]".format(self.string, key))

    def shape(self):
        return PartialLink(vars=self.vars, expr=sym.Expr("shape", (self.expr,)), links=self.links)

    def _flatten(self):
        return flatten_list([var._flatten() for var in self.vars]) + [self]
