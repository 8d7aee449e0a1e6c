"""
Graph objects: variables, symbolic links, probabilistic models.

Host-side counterpart of `brancher/variables.py` — same class names, constructor signatures, operator behaviour and
model bookkeeping as seen by user code — organised around what this engine needs:

* a ``PartialLink`` carries an explicit expression DAG (``symbolic.Expr``) instead of an opaque closure
  (`variables.py:977-1002`), so a (joint, posterior) pair is lowered *once* into a kernel program (`lowering.py`);
* graph objects hold no per-call state.  The reference memoises samples and visit-once flags on the variables themselves
  (``self.samples``, ``self._evaluated``, `variables.py:407-409,548-549,504-507`) and clears them with ``reset()``; here
  the compiled program is immutable and all per-call state lives in the engine workspace;
* arithmetic between graph objects is one table (``_OPERATORS``) installed on both ``Variable`` and ``PartialLink``;
  what a variable has been told about its data is one record (``_Observation``).

Numerical evaluation never happens in this module: ``_get_sample``, ``calculate_log_probability`` and
``estimate_log_model_evidence`` hand over to the native engine (`engine.py`), which raises if the HIP library or the GPU
is missing.
"""
from abc import ABC, abstractmethod
from collections.abc import Hashable, Iterable
import warnings

import numpy as np

from brancher_amd import distributions
from brancher_amd import symbolic as sym
from brancher_amd.modules import Parameter, ParameterModule
from brancher_amd.utilities import coerce_to_dtype, is_discrete, to_numpy


# ---------------------------------------------------------------------------------------------------------------------
#  arithmetic
# ---------------------------------------------------------------------------------------------------------------------
def _reflect_sub(link, other):        # other - link
    return -1 * (link - other)


def _reflect_div(link, other):        # other / link
    return (link / other) ** (-1)


# python operator -> (symbolic op, reflected form).  The reflected forms build the same expression shapes as the
# reference's (`variables.py:230-262`): b - a is -1 * (a - b), b / a is (a / b) ** -1, addition and multiplication commute.
_OPERATORS = {
    "add": ("add", lambda link, other: link + other),
    "sub": ("sub", _reflect_sub),
    "mul": ("mul", lambda link, other: link * other),
    "truediv": ("truediv", _reflect_div),
    "pow": ("pow", None),             # a number to the power of a variable is not defined by the reference either
}


def _with_operators(cls):
    """install +, -, *, /, ** (and their reflected forms) on a class that implements ``_apply_operator``"""
    def forward(op):
        return lambda self, other: self._apply_operator(other, op)

    def reflected(name, rule):
        if rule is None:
            def undefined(self, other):
                raise NotImplementedError("{!r} ** {}".format(other, type(self).__name__))
            return undefined
        return lambda self, other: rule(self, other)

    for name, (op, rule) in _OPERATORS.items():
        setattr(cls, "__{}__".format(name), forward(op))
        setattr(cls, "__r{}__".format(name), reflected(name, rule))
    cls.__neg__ = lambda self: -1 * self
    return cls


def _index_expression(key, strings_allowed):
    """the slice a ``x[key]`` link applies to a value laid out [samples, ...]: the sample axis is always kept whole.
    Integers and tuples of integers index behind it; any other hashable key (a string: the output name of a
    dictionary-valued link) passes through."""
    whole = slice(None, None, None)
    if isinstance(key, str):
        if strings_allowed:
            return key
    elif isinstance(key, Iterable):
        if strings_allowed or all(isinstance(k, int) for k in key):
            return (whole,) + tuple(key)
    elif isinstance(key, int) or strings_allowed:
        return (whole, key)
    if isinstance(key, Hashable):
        return key
    raise ValueError("cannot index a link with {!r}: expected integers or a hashable key".format(key))


# ---------------------------------------------------------------------------------------------------------------------
#  variables
# ---------------------------------------------------------------------------------------------------------------------
class BrancherClass(ABC):
    """Abstract superclass of variables, links and models (`variables.py:47-100`)."""

    @abstractmethod
    def _flatten(self):
        pass

    def flatten(self):
        return set(self._flatten())

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        """a torch function applied to a variable or a link — what happens inside a user callable wrapped by
        ``BrancherFunction(fn)`` (`functions.py:9-45`) when it is traced with symbolic arguments: ``torch.exp(x)`` becomes the
        link ``BF.exp(x)``, ``torch.add / sub / mul / div / pow`` the operators"""
        from brancher_amd import functions as BF
        name = getattr(func, "__name__", None)
        kwargs = kwargs or {}
        binary = {"add": "__add__", "sub": "__sub__", "mul": "__mul__", "div": "__truediv__", "true_divide": "__truediv__",
                  "pow": "__pow__"}
        if name in binary and len(args) == 2 and not kwargs:
            a, b = args
            if isinstance(a, BrancherClass):
                return getattr(a, binary[name])(b)
            return getattr(b, binary[name].replace("__", "__r", 1))(a)
        if name in ("neg", "negative") and len(args) == 1:
            return -args[0]
        if not name or name.startswith("_"):
            return NotImplemented
        return getattr(BF, name)(*args, **kwargs)

    def get_variable(self, var_name):
        # `variables.py:68-83`: a name -> variable table over the flattened graph.  Duplicate names are legal in the
        # reference (the README model has one) and the later one in name order wins; the lowering warns once per compile.
        by_name = {}
        for var in self._flatten():
            by_name[var.name] = var
        return by_name[var_name]


@_with_operators
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

    def __getitem__(self, key):
        # `variables.py:279-289`
        return PartialLink(vars={self}, expr=sym.Expr("getitem", (sym.variable(self),), _index_expression(key, True)),
                           links=set(), string="{}[{}]".format(self.name, key))

    def shape(self):
        return PartialLink(vars={self}, expr=sym.Expr("shape", (sym.variable(self),)), links=set())

    # ---- user-facing sampling: served by the native engine
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
        self.parents, self.ancestors = set(), set()
        self._type = "Deterministic"
        self._observed = is_observed
        self._value = coerce_to_dtype(data, is_observed)
        self.link = None
        self.learnable = bool(learnable) and not is_discrete(data)
        if learnable and not self.learnable:
            warnings.warn('Currently discrete parameters are not learnable. Learnable set to False')
        if self.learnable:
            self.link = ParameterModule(Parameter(self._value, name=name))

    @property
    def value(self):
        return self.link().numpy() if self.learnable else self._value

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


class _Observation:
    """What a random variable has been told about its data: nothing, a value (an array in the reference layout
    [datapoints, ...]), another random variable that supplies it (a minibatch drawn per iteration) — or only that it IS
    observed (``is_observed=True`` at construction, the value to follow)."""
    __slots__ = ("value", "source", "declared")

    def __init__(self, value=None, source=None, declared=False):
        self.value, self.source, self.declared = value, source, declared

    def __bool__(self):
        return self.declared or self.value is not None or self.source is not None


class RandomVariable(Variable):
    """A node with a distribution, parents and a link (`variables.py:384-622`)."""

    def __init__(self, distribution, name, parents, link):
        self.name = name
        self.distribution = distribution
        self.link = link
        self.parents = parents
        self.ancestors = None
        self._type = "Random"
        self._observation = _Observation()

    # the observation record under the attribute names of the reference (the lowering reads them)
    @property
    def _observed(self):
        return bool(self._observation)

    @property
    def _observed_value(self):
        return self._observation.value

    @property
    def has_observed_value(self):
        return self._observation.value is not None

    @property
    def dataset(self):
        return self._observation.source

    @property
    def has_random_dataset(self):
        return self._observation.source is not None

    @property
    def is_observed(self):
        return bool(self._observation)

    @property
    def value(self):
        if not self.has_observed_value:
            raise AttributeError("{!r} has no value: only observed random variables do".format(self.name))
        return self._observation.value

    def observe(self, data):
        # `variables.py:572-590`: a value, a DataFrame column of that name, or a random variable as the data source
        if isinstance(data, RandomVariable):
            self._observation = _Observation(source=data)
            return
        if type(data).__name__ == "DataFrame":
            from brancher_amd.pandas_interface import pandas_frame2value
            data = pandas_frame2value(data, self.name)
        self._observation = _Observation(value=coerce_to_dtype(data, is_observed=True))

    def unobserve(self):
        self._observation = _Observation()

    def _flatten(self):
        return sorted(list(self.ancestors) + [self], key=lambda v: v.name)

    def _get_sample(self, number_samples=1, resample=True, observed=False, input_values={}, differentiable=True):
        from brancher_amd import engine
        return engine.sample_variables([self], number_samples, observed=observed, input_values=input_values)

    def calculate_log_probability(self, input_values, reevaluate=True, for_gradient=False,
                                  include_parents=True, normalized=True):
        from brancher_amd import engine
        return engine.log_probability([self], input_values, include_parents=include_parents, reevaluate=reevaluate)


# ---------------------------------------------------------------------------------------------------------------------
#  models
# ---------------------------------------------------------------------------------------------------------------------
def _graph_closure(members):
    """the members of a model and everything they depend on, in name order"""
    found = set()
    for var in members:
        found.add(var)
        found.update(var.ancestors)
    return sorted(found, key=lambda v: v.name)


class ProbabilisticModel(BrancherClass):
    """A collection of variables (`variables.py:625-881`)."""

    def __init__(self, variables):
        for member in variables:
            if not isinstance(member, (RootVariable, RandomVariable, ProbabilisticModel)):
                raise ValueError("a probabilistic model is made of variables (or models), not of {}".format(type(member).__name__))
        self._input_variables = variables
        self.variables = self.flatten()
        self.posterior_model = None
        self.posterior_sampler = None
        self.is_transformed = False
        self.diagnostics = {}
        self._compiled = {}     # (posterior id, estimator) -> engine.CompiledELBO / CompiledDense / CompiledAmortized
        self.observed_submodel = self
        if any(not var.is_observed for var in variables):
            self.update_observed_submodel()

    def __str__(self):
        return str(self.model_summary)

    @property
    def model_summary(self):
        from brancher_amd.pandas_interface import reformat_model_summary
        members = list(self.flatten())
        return reformat_model_summary([[v._type, v.parents, v.is_observed] for v in members],
                                      [v.name for v in members], ["Distribution", "Parents", "Observed"])

    @property
    def is_observed(self):
        return all(var.is_observed for var in self._flatten())

    def _flatten(self):
        return _graph_closure(self._input_variables)

    def _observation_table(self, data):
        """{variable: value} from what ``observe`` accepts: a DataFrame (one column per variable name), a dictionary
        keyed by variables, or one keyed by variable names"""
        if type(data).__name__ == "DataFrame":
            from brancher_amd.pandas_interface import pandas_frame2value
            data = {column: pandas_frame2value(data, index=column) for column in data}
        if not isinstance(data, dict):
            raise ValueError("observe() takes a dictionary {variable or name: value} or a pandas DataFrame")
        keys = list(data)
        if all(isinstance(k, Variable) for k in keys):
            return dict(data)
        if all(isinstance(k, str) for k in keys):
            return {self.get_variable(name): value for name, value in data.items()}
        raise ValueError("observe(): the keys must be all variables or all variable names")

    def observe(self, data):
        # `variables.py:681-693`
        for var, value in self._observation_table(data).items():
            if isinstance(var, RandomVariable):
                var.observe(value)

    def update_observed_submodel(self):
        self.observed_submodel = ProbabilisticModel([var for var in self._flatten() if var.is_observed])

    def _as_posterior(self, part):
        if isinstance(part, Variable):
            part = ProbabilisticModel([part])
        if isinstance(part, ProbabilisticModel):
            return PosteriorModel(part, joint_model=self)
        raise ValueError("a sampler is a probabilistic model, a variable, or an iterable of those (got {})"
                         .format(type(part).__name__))

    def set_posterior_model(self, model, sampler=None):
        # `variables.py:703-716`
        self.posterior_model = PosteriorModel(posterior_model=model, joint_model=self)
        self._compiled = {}
        if not sampler:
            return
        if isinstance(sampler, (ProbabilisticModel, Variable)):
            self.posterior_sampler = self._as_posterior(sampler)
        elif isinstance(sampler, Iterable):
            self.posterior_sampler = [self._as_posterior(part) for part in sampler]
        else:
            self._as_posterior(sampler)        # raises

    def check_posterior_model(self):
        if not self.posterior_model:
            raise AttributeError("this model has no posterior yet: call set_posterior_model first")

    def reset(self):
        pass

    # ---- numerical entry points: all served by the native engine
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
        (`engine.importance_log_weights`); the max-shifted normalisation over samples is host arithmetic as in the
        reference.  Returns a numpy array [N, 1] (and the log normalisation when asked)."""
        from brancher_amd import engine
        log_p, log_q = engine.importance_log_weights(self, q_model, q_samples)
        log_w = (log_p - log_q).detach().cpu().numpy().reshape(-1, 1)
        shift = log_w.max()
        unnormalised = np.exp(log_w - shift)
        total = unnormalised.sum()
        if give_normalization:
            return unnormalised / total, np.log(total) + shift
        return unnormalised / total

    def estimate_log_model_evidence(self, number_samples, method="ELBO", input_values={},
                                    for_gradient=False, posterior_model=(), gradient_estimator=None):
        # `variables.py:843-870`
        if not posterior_model:
            self.check_posterior_model()
            posterior_model = self.posterior_model
        if method != "ELBO":
            raise NotImplementedError("only the ELBO estimate of the model evidence is implemented (got {!r})".format(method))
        from brancher_amd import engine
        return engine.estimate_elbo(self, posterior_model, number_samples,
                                    for_gradient=for_gradient, gradient_estimator=gradient_estimator)


class PosteriorModel(ProbabilisticModel):
    """`variables.py:884-907`.  ``_input_variables`` is the *set* of all flattened variables of the given model (roots
    included), exactly as the reference passes ``posterior_model.variables`` (`variables.py:894`)."""

    def __init__(self, posterior_model, joint_model):
        super().__init__(posterior_model.variables)
        self.posterior_model = None
        self.joint_model = joint_model
        self._is_trained = False

    @property
    def model_mapping(self):
        from brancher_amd.lowering import get_model_mapping
        return get_model_mapping(self, self.joint_model)


# ---------------------------------------------------------------------------------------------------------------------
#  links
# ---------------------------------------------------------------------------------------------------------------------
def var2link(var):
    """`variables.py:910-922`: anything that can stand in an expression becomes a PartialLink; other objects pass through"""
    if isinstance(var, PartialLink):
        return var
    if isinstance(var, Variable):
        return PartialLink(vars={var}, expr=sym.variable(var), links=set(), string=str(var))
    if sym.is_numeric_constant(var):
        return PartialLink(vars=set(), expr=sym.const(var), links=set(), string=str(var))
    if type(var).__module__.split(".")[0] == "torch" and hasattr(var, "detach"):
        return PartialLink(vars=set(), expr=sym.const(to_numpy(var)), links=set(), string="tensor")
    if isinstance(var, (tuple, list)) and all(isinstance(v, (Variable, PartialLink)) for v in var):
        parts = [var2link(v) for v in var]
        return PartialLink(vars=set().union(*[l.vars for l in parts]),
                           expr=sym.Expr("tuple", tuple(l.expr for l in parts)),
                           links=set().union(*[l.links for l in parts]), string=str(var))
    return var


@_with_operators
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
            raise TypeError("cannot combine a symbolic link with {!r}".format(other))
        return PartialLink(vars=self.vars | other.vars, expr=sym.binary(op, self.expr, other.expr),
                           links=self.links | other.links,
                           string="(" + str(self) + sym.BINARY_SYMBOLS[op] + str(other) + ")")

    def __getitem__(self, key):
        # `variables.py:1037-1053`
        return PartialLink(vars=self.vars, expr=sym.Expr("getitem", (self.expr,), _index_expression(key, False)),
                           links=self.links, string="{}[{}]".format(self.string, key))

    def shape(self):
        return PartialLink(vars=self.vars, expr=sym.Expr("shape", (self.expr,)), links=self.links)

    def _flatten(self):
        members = []
        for var in self.vars:
            members.extend(var._flatten())
        return members + [self]
