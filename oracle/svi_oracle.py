"""
ORACLE — test infrastructure, not product code.

CPU restatement of the reference's ELBO-gradient path, used ONLY by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker / baseline.
The product path (brancher_amd/) never imports this module.

What is restated (reference file:line in every function): the recursive ancestral sampling
of the posterior, the name-based q->p sample re-assignment, the visit-once log-probability
recursion, the semi-analytic entropy, the Pathwise and BlackBox estimators and the
optimisation loop of `brancher/inference.py`.  The arithmetic of the reference lives in an
un-vendored dependency — ``torch.distributions`` / ATen (`requirements.txt:1` pins only
``pytorch>=1.0.0``; this image has torch 2.10.0) — reached from `brancher/distributions.py:108,
122,124,166,180`; the oracle calls the same ``torch.distributions`` classes on CPU tensors, so
it shares the reference's numerics exactly.

Pinning: the reference's own tests hold no assertions or golden values, so the
oracle is pinned against outputs of the reference itself, generated in the build container by
`oracle/gen_golden.py` (which imports /root/reference) and committed under tests/golden/.
`tests/test_oracle_golden.py` checks this module against every one of them.

It walks the graph objects of brancher_amd (which mirror the reference's classes) with
per-call dictionaries, exactly like the reference, instead of using the compiled program —
so it also checks the lowering.
"""
import operator

import numpy as np
import torch
from torch import distributions as td

from brancher_amd.variables import RootVariable, RandomVariable
from brancher_amd import distributions as D
from brancher_amd import symbolic as sym

TORCHDIST = {
    D.DIST_NORMAL: td.normal.Normal, D.DIST_LOGNORMAL: td.log_normal.LogNormal,
    D.DIST_CAUCHY: td.cauchy.Cauchy, D.DIST_LAPLACE: td.laplace.Laplace, D.DIST_BETA: td.beta.Beta,
    D.DIST_BINOMIAL: td.binomial.Binomial, D.DIST_BERNOULLI: td.bernoulli.Bernoulli,
}
BINOPS = {"add": operator.add, "sub": operator.sub, "mul": operator.mul, "truediv": operator.truediv,
          "pow": operator.pow}


# ---- brancher/utilities.py restated -------------------------------------------------------
def sum_from_dim(t, dim_index):
    # utilities.py:119-126
    for dim in reversed(range(dim_index, t.dim())):
        t = t.sum(dim=dim)
    return t


def partial_broadcast(*args):
    # utilities.py:132-136
    s0 = max(x.shape[0] for x in args)
    s1 = max(x.shape[1] for x in args)
    return [x.expand((s0, s1) + tuple(x.shape[2:])) for x in args]


def broadcast_and_squeeze(*args):
    # utilities.py:143-148 + uniform_shapes 274-279
    if all(int(np.prod(v.shape[2:])) == 1 for v in args):
        args = [v.contiguous().view(tuple(v.shape[:2]) + (1, 1)) for v in args]
    max_len = max(len(a.shape) for a in args)
    args = [a.unsqueeze(len(a.shape)) if len(a.shape) == max_len - 1 else a for a in args]
    return torch.broadcast_tensors(*args)


def is_discrete(data):
    # utilities.py:31-32
    return type(data) in [list, set, tuple, dict, str]


def number_samples_and_datapoints(values):
    # utilities.py:189-207
    n_list, m_list = [], []
    for v in values.values():
        if torch.is_tensor(v):
            n_list.append(v.shape[0])
            m_list.append(v.shape[1])
    if not n_list:
        return None, None
    return max(n_list), max(m_list)


def tile_parameter(t, n):
    # utilities.py:257-266
    if t.shape[0] == n:
        return t
    if t.shape[0] == 1:
        return t.repeat(*([n] + [1] * (t.dim() - 1)))
    raise ValueError("The parameter cannot be broadcasted to the required number of samples")


def flatten_parent(v, n, m):
    # utilities.py:138-186 (tile_batch_dimensions + reshape_parent_value)
    v = v.expand((n, m) + tuple(v.shape[2:]))
    return v.contiguous().view((n * m,) + tuple(v.shape[2:]))


class _GivenBeta(torch.autograd.Function):
    """A Beta draw whose value is supplied, with the implicit-reparameterisation gradient of
    torch's `_Dirichlet` Function (torch/distributions/dirichlet.py:17-36, beta.py:85-86)."""

    @staticmethod
    def forward(ctx, concentration, x):
        ctx.save_for_backward(x, concentration)
        return x.clone()

    @staticmethod
    def backward(ctx, grad_output):
        x, concentration = ctx.saved_tensors
        total = concentration.sum(-1, True).expand_as(concentration)
        grad = torch._dirichlet_grad(x, concentration, total)
        return grad * (grad_output - (x * grad_output).sum(-1, True)), None


def _call(name):
    if name == "delta":
        return lambda x, y: (x == y).float()          # utilities.py:357-358
    if hasattr(torch, name):
        return getattr(torch, name)
    return getattr(torch.nn.functional, name)          # functions.py:50-62


class Oracle:
    def __init__(self, joint_model, posterior_model=None, dtype=torch.float32):
        self.p = joint_model
        self.q = posterior_model if posterior_model is not None else joint_model.posterior_model
        self.dtype = dtype
        self.minibatch = None  # {RandomIndices name: indices} for parity runs
        self.params = {}     # id(Parameter) -> torch leaf
        self.param_objs = {}
        for model in (self.q, self.p):
            for v in model.flatten():
                if isinstance(v, RootVariable) and v.learnable:
                    par = v.parameter
                    if id(par) not in self.params:
                        self.params[id(par)] = torch.tensor(par.numpy(), dtype=dtype, requires_grad=True)
                        self.param_objs[id(par)] = (v.name, par)
        # tensors of nn.Modules used as links (functions.py:15-20): optimised with the model's parameters through the
        # LinkConstructor (optimizers.py:36-49)
        self.module_links = []
        for model in (self.q, self.p):
            for v in model.flatten():
                link = getattr(v, "link", None)
                if link is None or not hasattr(link, "expressions"):
                    continue
                for l in link.expressions().values():
                    self._collect_module_links(getattr(l, "expr", None))
        for ml in self.module_links:
            for pname, par in ml.named.items():
                if id(par) not in self.params:
                    self.params[id(par)] = torch.tensor(par.numpy(), dtype=dtype, requires_grad=True)
                    self.param_objs[id(par)] = (par.name, par)
        # q->p mapping by name, built once here (reference: rebuilt per call, utilities.py:282-293)
        table = {v.name: v for v in self.q._flatten()}
        self.mapping = {}
        for p_var in self.p._flatten():
            if p_var.name in table:
                self.mapping[table[p_var.name]] = p_var

    def _collect_module_links(self, e):
        from brancher_amd.functions import ModuleLink
        if not isinstance(e, sym.Expr):
            return
        if e.op == "call" and isinstance(e.attr[0], ModuleLink) and e.attr[0] not in self.module_links:
            self.module_links.append(e.attr[0])
        for a in e.args:
            self._collect_module_links(a)

    # ------------------------------------------------------------------ parameters
    def named_parameters(self):
        return {name: self.params[i] for i, (name, _) in self.param_objs.items()}

    def set_parameters(self, by_name):
        with torch.no_grad():
            for i, (name, _) in self.param_objs.items():
                if name in by_name:
                    self.params[i].copy_(torch.as_tensor(np.asarray(by_name[name]), dtype=self.dtype)
                                         .reshape(self.params[i].shape))

    def zero_grad(self):
        for t in self.params.values():
            t.grad = None

    def root_value(self, root):
        # variables.py:352-356
        if root.learnable:
            return self.params[id(root.parameter)]
        if is_discrete(root.value):
            return root.value
        return torch.as_tensor(root.value, dtype=self.dtype)

    # ------------------------------------------------------------------ links
    def eval_expr(self, e, values):
        if e.op == "var":
            return values[e.attr]
        if e.op == "const":
            return e.attr                                           # variables.py:914-916
        if e.op in BINOPS:
            a, b = self.eval_expr(e.args[0], values), self.eval_expr(e.args[1], values)
            if isinstance(a, np.ndarray):
                a = torch.as_tensor(a, dtype=self.dtype)
            if isinstance(b, np.ndarray):
                b = torch.as_tensor(b, dtype=self.dtype)
            return BINOPS[e.op](a, b)                               # variables.py:995-1002
        if e.op == "call":
            fn, kwargs = e.attr
            f = _call(fn) if isinstance(fn, str) else fn
            if hasattr(fn, "named") and hasattr(fn, "module"):      # an nn.Module link: called on the value with THIS oracle's tensors
                mod, tensors = fn.module, {pname: self.params[id(par)] for pname, par in fn.named.items()}
                if self.dtype != torch.float32:
                    import copy
                    mod = copy.deepcopy(mod).to(self.dtype)
                f = lambda x, _m=mod, _t=tensors: torch.func.functional_call(_m, _t, (x,))      # noqa: E731
            args = [self.eval_expr(a, values) if isinstance(a, sym.Expr) else a for a in e.args]
            kw = {k: (self.eval_expr(v, values) if isinstance(v, sym.Expr) else v) for k, v in kwargs.items()}
            return f(*args, **kw)                                   # functions.py:34-38
        if e.op == "getitem":
            return self.eval_expr(e.args[0], values)[e.attr]
        if e.op == "tuple":
            return tuple(self.eval_expr(a, values) for a in e.args)
        if e.op == "shape":
            return self.eval_expr(e.args[0], values).shape
        raise NotImplementedError(e.op)

    def apply_link(self, var, parents_values):
        # variables.py:436-449 (discrete values — lists of indices — bypass the reshaping)
        n, m = number_samples_and_datapoints(parents_values)
        reshaped = {k: (v if is_discrete(v) else flatten_parent(v, n, m)) for k, v in parents_values.items()}
        out = {k: self.eval_expr(link.expr, reshaped) for k, link in var.link.expressions().items()}
        return {k: (v.view((n, m) + tuple(v.shape[1:])) if torch.is_tensor(v) else v) for k, v in out.items()}

    def is_det_node(self, v):
        return getattr(v, "_type", None) == "Deterministic node"

    # ------------------------------------------------------------------ sampling
    def sample_var(self, var, n, input_values, memo, noise, resample=False):
        """RootVariable._get_sample (variables.py:367-375) / RandomVariable._get_sample
        (variables.py:527-570).  `memo` plays the role of ``self.samples``."""
        if isinstance(var, RootVariable):
            value = input_values[var] if var in input_values else self.root_value(var)
            if is_discrete(value):
                return {var: value}                                       # variables.py:372-375
            return {var: tile_parameter(value, n)}
        if var in memo and not resample:
            return {var: memo[var]}
        if var in input_values:
            return {var: input_values[var]}
        parents_samples = {}
        for parent in var.parents:
            parents_samples.update(self.sample_var(parent, n, input_values, memo, noise, resample))
        params = self.apply_link(var, {p: parents_samples[p] for p in var.parents})
        sample = self.dist_sample(var, params, noise)
        memo[var] = sample
        out = dict(parents_samples)
        out[var] = sample
        return out

    def sample_posterior(self, n, noise=None):
        # ProbabilisticModel._get_sample (variables.py:732-742) over PosteriorModel's variables
        memo, joint = {}, {}
        for var in self.q._input_variables:
            joint.update(self.sample_var(var, n, {}, memo, noise, resample=False))
        return joint

    def dist_sample(self, var, params, noise):
        # distributions.py:70-75,111-124 (+ Deterministic :348-357)
        dist = var.distribution
        if dist.kind == D.DIST_DETERMINISTIC:
            return params["value"]
        keys = list(params.keys())
        vals = broadcast_and_squeeze(*[params[k] for k in keys])   # distributions.py:197-199
        params = dict(zip(keys, vals))
        tdist = TORCHDIST[dist.kind](**params)
        given = None if noise is None else noise.get(var.name)
        if given is None:
            if dist.has_differentiable_samples:
                return tdist.rsample()
            return tdist.sample()
        g = torch.as_tensor(np.asarray(given)).to(self.dtype)
        if dist.kind in (D.DIST_NORMAL, D.DIST_CAUCHY):
            return params["loc"] + g * params["scale"]              # torch normal.py:83-86, cauchy.py:77-80
        if dist.kind == D.DIST_LOGNORMAL:
            return torch.exp(params["loc"] + g * params["scale"])  # torch log_normal.py / ExpTransform
        if dist.kind == D.DIST_LAPLACE:
            return params["loc"] - params["scale"] * g.sign() * torch.log1p(-g.abs())   # torch laplace.py:83-86
        if dist.kind == D.DIST_BETA:
            conc = torch.stack([params["concentration1"], params["concentration0"]], -1)
            shape = torch.broadcast_shapes(conc.shape[:-1], g.shape)
            conc = conc.expand(tuple(shape) + (2,))
            g = g.expand(shape)
            x = torch.stack([g, 1.0 - g], -1)
            return _GivenBeta.apply(conc, x).select(-1, 0)          # torch beta.py:85-86
        return g.expand(torch.broadcast_shapes(g.shape, vals[0].shape)).clone()   # .sample(): no gradient path

    # ------------------------------------------------------------------ node statistics
    def parameters_from_input_values(self, var, input_values):
        # variables.py:451-468
        n = number_samples_and_datapoints(input_values)[0] if input_values else 1
        parents_values = {}
        for parent in var.parents:
            if parent in input_values:
                parents_values[parent] = input_values[parent]
        for parent in var.parents:
            if isinstance(parent, RootVariable) or self.is_det_node(parent):
                parents_values[parent] = self.sample_var(parent, n, input_values, {}, None, resample=True)[parent]
        return self.apply_link(var, parents_values)

    def empirical_draw(self, var, params):
        """EmpiricalDistribution._get_sample (distributions.py:410-462): minibatch without replacement;
        indices come from `self.minibatch[var.name]` when supplied (parity runs), else numpy's RNG."""
        dataset = params["dataset"]
        batch_size = var.distribution.batch_size
        if "indices" in params:
            indices = params["indices"]
        else:
            size = dataset.shape[1] if torch.is_tensor(dataset) else len(dataset)
            if self.minibatch is not None and var.name in self.minibatch:
                indices = [np.int64(i) for i in self.minibatch[var.name]]
            else:
                indices = list(np.random.choice(range(size), size=batch_size, replace=False))
        if torch.is_tensor(dataset):
            return dataset[:, [int(i) for i in indices], :]                 # is_observed branch (:455)
        return list(np.array(dataset)[[int(i) for i in indices]])

    def sample_observed(self, var, memo):
        """RandomVariable._get_sample(observed=True) (variables.py:548-570)."""
        if isinstance(var, RootVariable):
            value = self.root_value(var)
            return value if is_discrete(value) else tile_parameter(value, 1)
        if var in memo:
            return memo[var]
        if var.has_observed_value:
            memo[var] = torch.as_tensor(var._observed_value, dtype=self.dtype)
            return memo[var]
        target = var.dataset if var.has_random_dataset else var
        parents = {p: self.sample_observed(p, memo) for p in target.parents}
        params = self.apply_link(target, parents)
        if target.distribution.kind == D.DIST_EMPIRICAL:
            sample = self.empirical_draw(target, params)
        elif self.minibatch is not None and target.name in self.minibatch:
            # a variable observed BY FLAG only (examples/PopulationReceptiveFields.py:29): the reference draws it from its own
            # distribution once per evaluation (variables.py:553-565); parity runs hand the drawn value in, like minibatch rows
            sample = torch.as_tensor(np.asarray(self.minibatch[target.name]), dtype=self.dtype)
            while sample.dim() < 2 or sample.shape[0] != 1:
                sample = sample.unsqueeze(0)
        else:
            sample = self.dist_sample(target, params, None)
        memo[var] = sample
        return sample

    def dist_log_prob(self, var, x, params):
        # distributions.py:63-68,170-181; Implicit (Deterministic, Empirical) returns zeros (:226-227)
        dist = var.distribution
        if dist.kind in (D.DIST_DETERMINISTIC, D.DIST_EMPIRICAL):
            return torch.zeros((1, 1), dtype=self.dtype)
        if dist.kind == D.DIST_CATEGORICAL:
            # VectorDistribution preprocessing (distributions.py:257-266) + Categorical (:294-311)
            both = dict(params)
            both["x_data"] = x
            n, m = number_samples_and_datapoints(both)
            flat = {k: flatten_parent(v, n, m) for k, v in both.items()}
            flat = {k: v.contiguous().view(v.shape[0], int(np.prod(v.shape[1:]))) for k, v in flat.items()}
            xv = flat.pop("x_data")
            lp = td.categorical.Categorical(**flat).log_prob(xv[:, 0])
            return lp.contiguous().view(n, m)
        if dist.kind == D.DIST_MVNORMAL:
            # VectorDistribution preprocessing (distributions.py:257-269): value and the vector parameter `loc` are tiled to
            # [N*B-flattened rows, D]; the matrix parameter goes to torch as broadcast by the link; log_prob -> [N, B]
            # (torch multivariate_normal.py: -0.5 * (D log 2pi + |L^-1 (x - loc)|^2) - sum log diag L)
            both = dict(params)
            both["x_data"] = x
            n, m = number_samples_and_datapoints(both)
            flat = {k: flatten_parent(v, n, m) for k, v in both.items()}
            for k in ("loc", "x_data"):
                flat[k] = flat[k].contiguous().view(flat[k].shape[0], int(np.prod(flat[k].shape[1:])))
            xv = flat.pop("x_data")
            lp = td.multivariate_normal.MultivariateNormal(**flat).log_prob(xv)
            return lp.contiguous().view(n, m)
        keys = list(params.keys())
        vals = broadcast_and_squeeze(x, *[params[k] for k in keys])         # distributions.py:201-203
        x = vals[0]
        params = dict(zip(keys, vals[1:]))
        tdist = TORCHDIST[dist.kind](**params)
        return sum_from_dim(tdist.log_prob(x), 2)                            # utilities.py:128-129

    def dist_entropy(self, var, params):
        # distributions.py:91-96,155-168; Deterministic :381-390
        dist = var.distribution
        if dist.kind == D.DIST_DETERMINISTIC:
            return torch.zeros((1, 1, 1), dtype=self.dtype)
        keys = list(params.keys())
        vals = broadcast_and_squeeze(*[params[k] for k in keys])
        return TORCHDIST[dist.kind](**dict(zip(keys, vals))).entropy()

    def var_log_prob(self, var, input_values, evaluated, reevaluate=True, include_parents=True):
        """RootVariable.calculate_log_probability (variables.py:333-350) /
        RandomVariable.calculate_log_probability (variables.py:486-520)."""
        if isinstance(var, RootVariable):
            return torch.zeros((1, 1), dtype=self.dtype)
        if var in evaluated and not reevaluate:
            return 0.
        if var in input_values:
            value = input_values[var]
        elif var.is_observed and var.has_observed_value:
            value = torch.as_tensor(var._observed_value, dtype=self.dtype)
        elif self.is_det_node(var):
            value = None     # Implicit distribution ignores it
        else:
            raise AttributeError('RandomVariable has to be observed to receive value.')
        evaluated.add(var)
        params = self.parameters_from_input_values(var, input_values)
        lp = self.dist_log_prob(var, value, params)
        parents_lp = sum([self.var_log_prob(parent, input_values, evaluated, reevaluate) for parent in var.parents])
        if var.is_observed:
            lp = lp.sum(dim=1, keepdim=True)                                 # variables.py:513-514
        if torch.is_tensor(lp) and torch.is_tensor(parents_lp):
            lp, parents_lp = partial_broadcast(lp, parents_lp)
        return lp + parents_lp if include_parents else lp

    def model_log_prob(self, model, rv_values):
        # ProbabilisticModel.calculate_log_probability (variables.py:718-727)
        evaluated = set()
        return sum([self.var_log_prob(v, rv_values, evaluated, reevaluate=False) for v in model._input_variables])

    def var_entropy(self, var, input_values):
        # Variable._get_entropy (variables.py:156-162)
        if var.distribution.has_analytic_entropy:
            if isinstance(var, RootVariable):
                ent = torch.zeros((1, 1, 1), dtype=self.dtype)                # variables.py:362-365
            else:
                ent = self.dist_entropy(var, self.parameters_from_input_values(var, input_values))
            return sum_from_dim(ent, 2)
        return -self.var_log_prob(var, input_values, set(), reevaluate=True, include_parents=False)

    def var_mean(self, var, input_values):
        # RootVariable._get_statistic returns the value (variables.py:362-365); RandomVariable._get_statistic evaluates
        # the parameters on the input values and asks the distribution (:522-525; Deterministic mean = value,
        # distributions.py:359-368; torch distributions' .mean otherwise, :137)
        if isinstance(var, RootVariable):
            return self.root_value(var)
        params = self.parameters_from_input_values(var, input_values)
        if var.distribution.kind == D.DIST_DETERMINISTIC:
            return params["value"]
        keys = list(params.keys())
        vals = broadcast_and_squeeze(*[params[k] for k in keys])
        return TORCHDIST[var.distribution.kind](**dict(zip(keys, vals))).mean

    def posterior_entropy(self, samples):
        # ProbabilisticModel._get_entropy (variables.py:744-749)
        ents = [self.var_entropy(v, samples) for v in self.q.variables]
        return sum([sum_from_dim(e, 2) for e in ents])

    def empirical_samples(self):
        # observed_submodel._get_sample(1, observed=True) (variables.py:849, 548-570): every observed
        # variable of the joint model, one shared memo so that minibatch indices are drawn once
        memo = {}
        out = {}
        for v in self.p._flatten():
            if isinstance(v, RandomVariable) and v.is_observed:
                out[v] = self.sample_observed(v, memo)
        return out

    def p_log_prob_from_q_samples(self, q_samples, empirical):
        # get_p_log_probabilities_from_q_samples (variables.py:814-819) + reassign_samples (utilities.py:296-309)
        p_samples = {self.mapping[k]: v for k, v in q_samples.items() if k in self.mapping}
        p_samples.update(empirical)
        return self.model_log_prob(self.p, p_samples)

    # ------------------------------------------------------------------ estimators
    def function(self, samples, empirical):
        # the closure of estimate_log_model_evidence (variables.py:851-855)
        return self.p_log_prob_from_q_samples(samples, empirical) + self.posterior_entropy(samples)

    def elbo(self, n, estimator="pathwise", noise=None, return_parts=False):
        """PathwiseDerivativeEstimator (gradient_estimators.py:39-44) /
        BlackBoxEstimator (gradient_estimators.py:29-36).  Returns the estimator value
        (loss = -value, inference.py:141)."""
        empirical = self.empirical_samples()
        samples = self.sample_posterior(n, noise)
        samples.update(empirical)
        if estimator == "pathwise":
            f = self.function(samples, empirical)
            value = f.mean()
            lq = None
        elif estimator == "blackbox":
            lq = self.model_log_prob(self.q, samples)
            f = self.function(samples, empirical)
            value = (lq * f.detach() + self.function(samples, empirical)).mean()
        elif estimator == "taylor1":
            # Taylor1Estimator (gradient_estimators.py:47-56): f at the analytic means of the sampler's unobserved
            # variables given the SAMPLED values (Variable._get_mean, variables.py:85-86 -> _get_statistic :522-525 ->
            # Distribution.get_mean, distributions.py:77-82,126-139); everything else keeps its sampled value
            means = {v: self.var_mean(v, samples) for v in self.q.variables if not v.is_observed}
            for k, v in samples.items():
                if k not in means:
                    means[k] = v
            f = self.function(means, empirical)
            value = f.mean()
            lq = None
        elif callable(estimator):
            # a user-defined GradientEstimator (gradient_estimators.py:17-26): any scalar of the per-sample f and log q of
            # one draw, differentiated by autograd — `estimator(f, log_q)` restates its __call__ on the two tensors
            lq = self.model_log_prob(self.q, samples)
            f = self.function(samples, empirical)
            value = estimator(f, lq)
        else:
            raise ValueError(estimator)
        if return_parts:
            named = {k.name: v for k, v in samples.items() if isinstance(k, RandomVariable) and torch.is_tensor(v)}
            return value, f, lq, named
        return value

    def log_densities(self, named_q_samples):
        """log p(z, y) and log q(z) at SUPPLIED posterior samples {variable name: [N, ...]}: what
        `ProbabilisticModel.get_importance_weights` (variables.py:821-841) computes its weights from —
        `get_p_log_probabilities_from_q_samples` (:814-819) and the posterior's `calculate_log_probability` (:718-727)."""
        by_name = {v.name: v for v in self.q._flatten()}
        samples = {by_name[name]: torch.as_tensor(np.asarray(value), dtype=self.dtype) for name, value in named_q_samples.items()}
        empirical = self.empirical_samples()
        with torch.no_grad():
            lp = self.p_log_prob_from_q_samples(samples, empirical)
            both = dict(samples)
            both.update(empirical)
            lq = self.model_log_prob(self.q, both)
        return lp.reshape(-1).numpy().copy(), lq.reshape(-1).numpy().copy()

    def loss_and_grads(self, n, estimator="pathwise", noise=None, minibatch=None):
        self.minibatch = minibatch
        self.zero_grad()
        value, f, lq, samples = self.elbo(n, estimator, noise, return_parts=True)
        loss = -value
        loss.backward()
        grads = {}
        for i, (name, _) in self.param_objs.items():
            g = self.params[i].grad
            grads[name] = None if g is None else g.detach().numpy().copy()
        return dict(loss=float(loss.detach()), f=f.detach().numpy().copy(),
                    lq=None if lq is None else lq.detach().numpy().copy(), grads=grads,
                    samples={k: v.detach().numpy().copy() for k, v in samples.items() if torch.is_tensor(v)})

    # ------------------------------------------------------------------ the loop
    def make_optimizers(self, optimizer="SGD", **opt_params):
        """ProbabilisticOptimizer per model (optimizers.py:53-67, inference.py:79-89):
        posterior first, then the joint model if it owns parameters."""
        groups = [[], []]
        q_roots = {v for v in self.q.variables if isinstance(v, RootVariable)}
        for i, (name, par) in self.param_objs.items():
            in_q = any(v.learnable and v.parameter is par for v in q_roots)
            groups[0 if in_q else 1].append(self.params[i])
        return [getattr(torch.optim, optimizer)(g, **opt_params) for g in groups if g]

    def train(self, iterations, n, optimizer="SGD", estimator="pathwise", noise_seq=None,
              pretraining_iterations=0, minibatch_seq=None, **opt_params):
        """inference.py:95-108 (one loss entry per iteration; the reference appends twice)."""
        opts = self.make_optimizers(optimizer, **opt_params)
        losses = []
        for it in range(iterations):
            noise = None if noise_seq is None else noise_seq[it]
            self.minibatch = None if minibatch_seq is None else minibatch_seq[it]
            loss = -self.elbo(n, estimator, noise)
            if torch.isfinite(loss.detach()).all().item():
                for o in opts:
                    o.zero_grad()
                loss.backward()
                opts[0].step()
                if it > pretraining_iterations:
                    for o in opts[1:]:
                        o.step()
            losses.append(float(loss.detach()))
        return np.array(losses, dtype=np.float32)
