"""
Lowering: (joint model, posterior model, estimator)  ->  immutable kernel program.

What the reference does *per ELBO evaluation* in Python — the recursive sampling walk
(`brancher/variables.py:527-570,732-742`), the name-based q->p re-mapping
(`brancher/utilities.py:282-309`, rebuilt every call), the visit-once log-probability
recursion (`variables.py:486-520,718-727`) and the entropy pass (`variables.py:744-749,
156-162`) — is resolved here *once* into a flat program (include/bsvi.h):

  uniform table   lane-uniform values U[k] = a + b*g(theta_i | const_i): every parameter
                  transform of `geometric_ranges.py` is hoisted here, so the per-sample code
                  never evaluates softplus/sigmoid of a parameter, and the gradient
                  reduction over samples happens on U, not on theta
  records         one per node evaluation, in dependency order: q nodes (SAMPLE +
                  ENTROPY [+ LOGP for the score term]) then p nodes (LOGP)
  micro-ops       the link expressions, register-allocated (SSA, <= BSVI_NUM_REGS)

Reference semantics reproduced on purpose (all probed against the reference, see
tests/golden and DESIGN.md §2):
  * a p-variable (root, deterministic or random) takes the value of the q-variable with
    the same *name* — including auto-created roots such as ``x3_scale``, so a prior's
    numeric parameters are silently replaced by the posterior's learnable ones when both
    use the same variable name (`utilities.py:282-309` + `variables.py:367-371,463-466`);
  * observed values override that mapping (`variables.py:817`);
  * ELBO = mean over the broadcast [N, B] of  log p + sum_v H_v  with analytic entropies
    where ``has_analytic_entropy`` else ``-log q_v`` (`variables.py:851-855,156-162`);
  * observed nodes are summed over the datapoint axis, others are not (`variables.py:513-514`).
"""
import warnings

import os

import math

import numpy as np

from brancher_amd import distributions as D
from brancher_amd import symbolic as sym
from brancher_amd.utilities import canonical_elem_shape, broadcast_shapes3, is_discrete
from brancher_amd.variables import RootVariable, RandomVariable, ProbabilisticModel

# ---- mirror of include/bsvi.h (tests/test_lowering_abi.py checks the two stay in sync) -----
OP = dict(NOP=0, NAFF=1, NODE=2, BIN=3, UN=4, REC_BEGIN=5, REC_END=6)
R_SINK = 1     # record flag: complete (forward and reverse) in the forward sweep
R_NOALIAS = 2  # instruction flag: no two operands share an adjoint cell (adjoint updates may be batched)
F_SAMPLE, F_ENT, F_LOGP, F_WF, F_GIVEN = 1, 2, 4, 8, 16
K_NONE, K_U, K_Z, K_OBS = 0, 1, 2, 3
BINOP = dict(add=0, sub=1, mul=2, truediv=3, pow=4, delta=5)
UNOP = dict(copy=0, neg=1, exp=2, log=3, sqrt=4, sin=5, cos=6, tanh=7, abs=8, sigmoid=9, softplus=10,
            relu=11, reciprocal=12, log1p=13, expm1=14, square=15, p2l=16, powi=17)
UT = dict(identity=0, softplus=1, sigmoid=2, exp=3, log=4, tanh=5, sqrt=6, square=7)
# "importance" is an evaluation program (no estimator of its own): the posterior's nodes take their values
# from the caller and only accumulate log q; the kernel's value is then log p(z, y) and its second
# per-sample output log q(z) — `ProbabilisticModel.get_importance_weights`, variables.py:821-841
# "taylor1" (`gradient_estimators.py:47-56`) is the pathwise accumulation of f evaluated at the posterior's analytic
# means given the sampled parents: a different PROGRAM for the same kernel estimator (see _Lowering.mean_value)
EST = dict(pathwise=0, blackbox=1, importance=0, taylor1=0)

UNARY_CALLS = set(UNOP) - {"copy", "powi"}

UNIFORM_DTYPE = np.dtype([("src", "<u4"), ("transform", "u1"), ("is_param", "u1"), ("reserved", "<u2"),
                          ("a", "<f4"), ("b", "<f4")])
RECORD_DTYPE = np.dtype([("code_begin", "<u4"), ("code_end", "<u4"), ("n_elems", "<u4"), ("temp_base", "<u4"),
                         ("n_temps", "<u4"), ("flags", "<u4")])


K_UCONST = 4   # lowering-internal: a uniform entry sourced from the constant buffer


def operand(kind, index=0, stride=0):
    """symbolic operand; encoded by _Lowering.finish() once the table sizes are known"""
    return (kind, int(index), int(stride))


def encode_operand(opnd, n_uniform_param, n_uniform):
    """include/bsvi.h operand word: byte offset | walks<<30 | per_lane<<31.
    U / OBS operands address the uniform region of LDS (4 bytes per entry, observed data behind
    the uniform table); Z operands address the lane's own slot row, where value and adjoint
    of a slot are interleaved (8 bytes per slot)."""
    kind, index, stride = opnd
    if kind == K_NONE:
        return 0
    if kind == K_U:
        off, per_lane = index * 4, 0
    elif kind == K_UCONST:
        off, per_lane = (n_uniform_param + index) * 4, 0
    elif kind == K_OBS:
        off, per_lane = (n_uniform + index) * 4, 0
    else:
        off, per_lane = index * 8, 1
    assert off < (1 << 30)
    return off | (stride << 30) | (per_lane << 31)


class LoweringError(NotImplementedError):
    """The model uses something the fused kernel cannot execute (yet)."""


def get_model_mapping(source_model, target_model):
    """q-variable -> p-variable by name (`brancher/utilities.py:282-293`)."""
    mapping = {}
    table = {v.name: v for v in source_model._flatten()}
    for p_var in target_model._flatten():
        if p_var.name in table:
            mapping[table[p_var.name]] = p_var
    return mapping


def _f32(x):
    return np.float32(x)


def _fbits(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


# ==========================================================================================
#  intermediate representation of link expressions with the model context resolved
# ==========================================================================================
class IR:
    __slots__ = ("op", "args", "attr", "shape", "key", "has_z")

    def __init__(self, op, args, attr, shape, key, has_z):
        self.op, self.args, self.attr, self.shape, self.key, self.has_z = op, args, attr, shape, key, has_z

    def __repr__(self):
        return "IR(%s%s)" % (self.op, "" if not self.args else "," + ",".join(a.op for a in self.args))


def _split_axis(rec_shape, leaf_shapes):
    """Partial broadcasting.  An instruction's operand either stays put or walks the record's element loop
    one element at a time, so inside ONE record every operand must be constant or contiguous.  A node of
    canonical shape (B, D1, D2) whose operands broadcast over some axes only (a [1, D1, 1] latent inside a
    [B, D1, 1] likelihood: the same vector for every datapoint) is therefore cut into one record per index of
    its first k axes; over the remaining axes every operand is then either full (walks) or of extent 1 (fixed).
    Returns the smallest such k (0 = one record over everything, the common case)."""
    rec_shape = tuple(rec_shape)
    for k in range(0, 3):
        inner = [i for i in range(k, 3) if rec_shape[i] > 1]
        ok = True
        for s in leaf_shapes:
            full = all(s[i] == rec_shape[i] for i in inner)
            flat = all(s[i] == 1 for i in inner)
            if not (full or flat):
                ok = False
                break
        if ok:
            return k
    return 3


class MinibatchObs:
    """Observed data that is a MINIBATCH: `batch` rows of `dataset` [DS, ...] per evaluation (an observed EmpiricalVariable,
    `standard_variables.py:71-96`).  Stands where an observed variable stands in the IR (`_observed_value`: the placeholder the
    observation buffer is laid out from); `indices` is the RandomIndices variable it shares with others, or None (its own draw)."""

    def __init__(self, source, dataset, batch, indices):
        self.source, self.name, self.dataset, self.batch, self.indices = source, source.name, dataset, batch, indices
        if batch > dataset.shape[0]:
            raise LoweringError("batch_size %d of %r exceeds its dataset (%d rows)" % (batch, source.name, dataset.shape[0]))
        self._observed_value = np.zeros((1, batch) + dataset.shape[1:], dtype=np.float32)      # [1, B, ...] like an observed value (utilities.py:226-232)
        self.is_observed = True


class _SurrogateTerm:
    """what the emission reads of a model term that is a LINEAR surrogate row (coefficient x partner): the reduce node's
    log-probability of its drawn data"""
    is_observed, b_axis = True, 1

    def __init__(self, name):
        self.name, self.distribution = name, D.LinearSurrogate()


class DrawnObs:
    """A variable that is observed BY FLAG only (`is_observed=True`, never given a value): the reference draws it from its own
    distribution ONCE per evaluation and every Monte-Carlo sample shares the draw (`variables.py:849`, `:553-565`).  A Normal whose
    mean / scale [B, D1, D2] are constants of the model; stands where an observed variable stands in the IR."""

    def __init__(self, var, mean, scale):
        self.var, self.name, self.mean, self.scale, self.consumed = var, var.name, mean, scale, False
        self._observed_value = np.zeros((1,) + mean.shape, dtype=np.float32)
        self.is_observed = True


class ExternalReduce:
    """A reduction evaluated by the reduce node of the library (lowering.reduce_external): what `native.reduce_desc` turns into a
    bsvi_reduce_desc, and where its rows live in the program's noise tensor."""
    kind = "reduce"
    name = code = mats = uniform_inputs = slot_inputs = data_mean = data_scale = drawn_name = logp_node = None
    rows = cols = n_data = row0 = n_rows_out = 0
    drawn, weight = False, 0.0


class SlotInfo:
    def __init__(self, var, base, shape, dist):
        self.var, self.base, self.shape, self.dist = var, base, shape, dist
        self.size = _nelem(shape)
        self.name = var.name


class Program:
    """Everything the engine needs to run a compiled (joint, posterior) pair."""

    def __init__(self):
        self.estimator = "pathwise"
        self.uniform = None
        self.records = None
        self.code = None
        self.consts = None
        self.obs = None
        self.n_params = 0
        self.n_slots = 0
        self.n_noise = 0
        self.n_uniform_grad = 0
        self.param_uniform_ptr = None
        self.param_uniform_idx = None
        self.parameters = []        # [(Parameter, offset, size, group)]
        self.param_active = None    # uint8 [n_params]: referenced by the program
        self.param_group = None     # uint8 [n_params]: 0 posterior, 1 joint model
        self.slots = {}             # q variable -> SlotInfo
        self.slot_by_name = {}
        self.bmax = 1
        self.n_derived = 0
        self.n_temps = 0
        self.op_count = 0

    def initial_params(self):
        theta = np.zeros(self.n_params, dtype=np.float32)
        for p, off, size, _ in self.parameters:
            theta[off:off + size] = p.numpy().reshape(-1)
        return theta

    def noise_rows(self, name):
        s = self.slot_by_name[name]
        return s.base, s.size, s.shape

    def summary(self):
        return dict(n_params=self.n_params, n_slots=self.n_slots, n_uniform=len(self.uniform),
                    n_uniform_grad=self.n_uniform_grad, n_records=len(self.records), n_code=len(self.code),
                    n_latent=self.n_noise, n_derived=self.n_derived, n_temps=self.n_temps, bmax=self.bmax,
                    estimator=self.estimator)


class _ModuleTensor:
    """One tensor of a ModuleLink seen by the program as a learnable root: what `uniform_entries` / `group_of` read of a
    RootVariable.  Never sampled, never observed; belongs to the joint model's parameter group (`optimizers.py:36-49`)."""
    learnable = True
    value = None

    def __init__(self, link, pname):
        self.link, self.pname = link, pname
        self.parameter = link.named[pname]
        self.name = "%s.%s" % (link.name, pname)


class _Lowering:
    def __init__(self, joint, posterior, estimator):
        if estimator not in EST:
            raise ValueError("unknown gradient estimator %r" % (estimator,))
        self.joint, self.posterior, self.estimator = joint, posterior, estimator
        self.ir_cache = {}
        self.q_by_name = {}
        self.slots = {}
        self.n_slots = 0
        self.param_index = {}      # id(Parameter) -> offset
        self.parameters = []
        self.n_params = 0
        self.consts = []           # list of float arrays
        self.n_consts = 0
        self.const_index = {}      # key -> offset
        self.obs = []
        self.n_obs = 0
        self.obs_index = {}        # id(var) -> (offset, shape)
        self.minibatch_obs = {}    # id(EmpiricalVariable) -> MinibatchObs: observations that are a minibatch of a dataset
        self.drawn_obs = {}        # id(variable observed by flag only) -> DrawnObs: drawn once per evaluation
        self.uni_param = []        # provisional uniform entries (param-sourced)
        self.uni_const = []
        self.uni_index = {}        # (kind, id/ key, transform, a, b) -> (is_param, local k0)
        self.module_roots = {}     # (id(ModuleLink), tensor name) -> _ModuleTensor
        self.code = []             # list of instructions with symbolic operands
        self.records = []
        self.derived = {}          # IR key -> derived slot base
        self.use_count = {}
        self.n_latent = 0
        self.n_derived = 0
        self.temp_base = 0
        self.max_temps = 0
        self.view_rank, self.view_memo, self.elem_memo, self.view_terms = {}, {}, {}, 0
        self.slot_rank = {}
        self.mean_entropy = {}      # Taylor1: q variable -> its parameters rebuilt on the parents' means (entropy-only record)
        self.externals, self.pseudo_q, self.external_mode = [], [], "inject"
        self.mean_q = []            # Taylor1: (pseudo variable, [mean expression, 0], shape) — a vector value's mean per sample, as rows of the draw

    # ---------------------------------------------------------------- IR construction
    def mk(self, op, args=(), attr=None, shape=None):
        if op in ("root", "z", "obs"):
            key = (op, id(attr))
        elif op == "elem":
            key = (op, int(attr)) + tuple(a.key for a in args)
        elif op == "imm":
            key = (op, float(attr))
        elif op == "carr":
            key = (op, attr.shape, attr.tobytes())
        else:
            key = (op, attr if not isinstance(attr, np.ndarray) else None) + tuple(a.key for a in args)
        hit = self.ir_cache.get(key)
        if hit is not None:
            return hit
        if shape is None:
            shape = broadcast_shapes3(*[a.shape for a in args]) if args else (1, 1, 1)
        has_z = (op == "z") or any(a.has_z for a in args)
        node = IR(op, tuple(args), attr, tuple(shape), key, has_z)
        self.ir_cache[key] = node
        return node

    def root_shape(self, var):
        v = var.value if not var.learnable else var.parameter.numpy()
        if is_discrete(v):
            raise LoweringError("discrete root value %r cannot enter the fused kernel" % (var.name,))
        return canonical_elem_shape(v.shape[1:])

    def from_expr(self, e, ctx):
        """sym.Expr -> IR with variables replaced through ctx (a function Variable -> IR)."""
        if e.op == "var":
            return ctx(e.attr)
        if e.op == "const":
            v = e.attr
            if isinstance(v, np.ndarray) and v.size > 1:
                arr = np.ascontiguousarray(v, dtype=np.float32)
                return self.mk("carr", (), arr, canonical_elem_shape((1,) + arr.shape))
            return self.mk("imm", (), float(np.asarray(v).reshape(-1)[0]) if isinstance(v, np.ndarray) else float(v))
        if e.op in sym.BINARY_OPS:
            a, b = self.from_expr(e.args[0], ctx), self.from_expr(e.args[1], ctx)
            return self.ranked(self.mk(e.op, (a, b)), self.check_ranks((a, b), e.op))
        if e.op == "getitem":
            if isinstance(e.attr, str):
                raise LoweringError("named outputs (%r) exist for network links only (amortised path)" % (e.attr,))
            return self.view_index(self.from_expr(e.args[0], ctx), e.attr)
        if e.op == "call":
            fn, kwargs = e.attr
            if not isinstance(fn, str):
                from brancher_amd.functions import ModuleLink
                if isinstance(fn, ModuleLink) and len(e.args) == 1 and isinstance(e.args[0], sym.Expr) and not kwargs:
                    return self.module_call(fn, self.from_expr(e.args[0], ctx))
                raise LoweringError("user callables / nn.Modules inside links are not lowered to the fused "
                                    "kernel yet: %r" % (fn,))
            if fn in ("sum", "transpose") and e.args:
                return self.view_call(fn, self.from_expr(e.args[0], ctx), list(e.args[1:]), kwargs)
            if fn == "matmul" and len(e.args) == 2 and not kwargs and all(isinstance(a, sym.Expr) for a in e.args):
                # BF.matmul(A, x) (`functions.py:50-62` wraps torch.matmul) with x a COLUMN [c, 1] — a linear predictor over a handful of
                # weights, the minibatched Bayesian linear regression next to `examples/minibatch_logistic_regression.py:13-51` —
                # unrolled through the axis views: sum_c A[r][c] x[c] = BF.sum(A * BF.transpose(x, 1, 2), dim=2, keepdim=True).  (Real
                # matrix products — a 10 x 784 weight matrix per sample — belong to the dense path: kMaxViewTerms refuses them here.)
                a, b = self.from_expr(e.args[0], ctx), self.from_expr(e.args[1], ctx)
                if b.shape[2] != 1 or a.shape[2] != b.shape[1]:
                    raise LoweringError("BF.matmul on the scalar path takes a matrix [r, c] and a column [c, 1] (got %r and %r)"
                                        % (a.shape[1:], b.shape[1:]))
                bt = self.view_call("transpose", b, [1, 2], {})
                prod = self.ranked(self.mk("mul", (a, bt)), self.check_ranks((a, bt), "BF.matmul"))
                return self.view_call("sum", prod, [], dict(dim=2, keepdim=True))
            if kwargs:
                raise LoweringError("keyword arguments of BF.%s are not supported by the fused kernel" % fn)
            args = [self.from_expr(a, ctx) if isinstance(a, sym.Expr) else self.mk("imm", (), float(a))
                    for a in e.args]
            rank = self.check_ranks(args, "BF." + fn)
            if fn in UNARY_CALLS and len(args) == 1:
                return self.ranked(self.mk("call:" + fn, (args[0],)), rank)
            if fn in ("delta",) and len(args) == 2:
                return self.ranked(self.mk("delta", tuple(args)), rank)
            if fn in ("add", "sub", "mul", "div", "true_divide", "pow") and len(args) == 2:
                return self.ranked(self.mk({"div": "truediv", "true_divide": "truediv"}.get(fn, fn), tuple(args)), rank)
            raise LoweringError("BF.%s is not in the fused kernel's op set" % fn)
        raise LoweringError("link expression node %r is not supported by the fused kernel" % (e.op,))

    # ---------------------------------------------------------------- axis views: sum / transpose / [...]
    # Inside a link the reference lays a value out [samples x datapoints, d1, d2] (`variables.py:436-449`,
    # `utilities.py:179-186`) and hands it to torch.sum / torch.transpose (`functions.py:50-62`) or indexes it behind the
    # first axis (`variables.py:279-289`): axis 1 / 2 of the link tensor are the canonical element axes d1 / d2 here, and
    # the result is viewed back as [samples, datapoints, ...].  The fused kernel walks an element loop in which every operand
    # is fixed or contiguous, so these are not instructions: a view node stays symbolic in the IR and is resolved PER OUTPUT
    # ELEMENT into scalar expressions over single elements of the leaves (`element_of`); a node whose parameters contain
    # views is emitted as one scalar term per element (`split_elements`).  A reduction over K elements is K-1 adds of the
    # unrolled element expressions: meant for short vectors (a linear predictor over a handful of weights) — K x E beyond
    # kMaxViewTerms is refused (the dense path exists for real matrix products).
    kMaxViewTerms = 4096

    @staticmethod
    def static_int(x, what):
        if isinstance(x, sym.Expr):
            if x.op != "const":
                raise LoweringError("%s must be a constant" % what)
            x = x.attr
        a = np.asarray(x).reshape(-1)
        if a.size != 1 or float(a[0]) != int(a[0]):
            raise LoweringError("%s must be an integer constant" % what)
        return int(a[0])

    def rank_of(self, node):
        """rank of the link tensor a node stands for: 3 ([rows, d1, d2]) unless an integer index or a sum without
        keepdim dropped an axis, or the leaf it derives from has fewer axes: inside a link the reference hands torch the
        value as [samples x datapoints] + shape[2:] (`variables.py:436-449`, `utilities.py:179-186`), and an unobserved
        1-D array (d,) is stored [1, 1, d] (`utilities.py:236`) — rank 2, so `dim=-1` / `dim=1` is its d axis and `dim=2`
        does not exist"""
        return self.view_rank.get(node.key, 3)

    def ranked(self, node, rank):
        if rank != 3:
            self.view_rank[node.key] = rank
        return node

    def module_call(self, link, x):
        """`BrancherFunction(nn.Module)(x)` on the scalar path (`brancher/functions.py:15-41`: the module is called on the value,
        its tensors are optimised with the model's other parameters through the LinkConstructor, `optimizers.py:36-49`).  A small
        MLP — `nn.Linear`, or an `nn.Sequential` of `nn.Linear` and Tanh / ReLU / Sigmoid / Softplus — acting on the LAST axis of
        its input is unrolled into the per-sample program: every weight is a learnable uniform entry (its gradient one position
        of the reduction), every unit a chain of multiply-adds.  The input is a scalar or a vector along the last axis; several
        output units come back as a `vstack` view along that axis."""
        import torch.nn as nn
        mod = link.module
        stages = list(mod.children()) if isinstance(mod, nn.Sequential) else [mod]
        prefixes = [str(i) + "." for i in range(len(stages))] if isinstance(mod, nn.Sequential) else [""]
        if x.shape[0] != 1 or x.shape[1] != 1:
            raise LoweringError("module link %s: the input must be a scalar or a vector along its last axis, got element shape %r"
                                % (link.name, (x.shape,)))
        units = [self.element_of(x, (0, 0, j)) for j in range(x.shape[2])]
        acts = {nn.Tanh: "tanh", nn.ReLU: "relu", nn.Sigmoid: "sigmoid", nn.Softplus: "softplus"}
        for stage, prefix in zip(stages, prefixes):
            if isinstance(stage, nn.Linear):
                if stage.in_features != len(units):
                    raise LoweringError("module link %s: Linear(%d, %d) applied to %d values" % (link.name, stage.in_features,
                                                                                              stage.out_features, len(units)))
                w = self.module_root(link, prefix + "weight", (1, stage.out_features, stage.in_features))
                b = self.module_root(link, prefix + "bias", (1, 1, stage.out_features)) if stage.bias is not None else None
                out = []
                for k in range(stage.out_features):
                    acc = None
                    for j, u in enumerate(units):
                        t = self.mk("mul", (self.element_of(w, (0, k, j)), u))
                        acc = t if acc is None else self.mk("add", (acc, t))      # (torch's addmm adds the products in this order)
                    if b is not None:
                        acc = self.mk("add", (acc, self.element_of(b, (0, 0, k))))
                    out.append(acc)
                units = out
            elif type(stage) in acts:
                if isinstance(stage, nn.Softplus) and (stage.beta != 1 or stage.threshold != 20):
                    raise LoweringError("module link %s: Softplus with non-default beta / threshold" % link.name)
                units = [self.mk("call:" + acts[type(stage)], (u,)) for u in units]
            elif isinstance(stage, nn.Identity):
                pass
            else:
                raise LoweringError("module link %s: %s is not lowered on the scalar path (Linear, Tanh, ReLU, Sigmoid, Softplus)"
                                    % (link.name, type(stage).__name__))
        if len(units) == 1:
            return units[0]
        # several output units (round 6): a VIEW whose element j is unit j — consumers resolve it per element like BF.sum / x[...]
        # (a model term over it becomes one scalar term per unit: split_elements)
        return self.ranked(self.mk("vstack", tuple(units), None, (1, 1, len(units))), self.rank_of(x))

    def module_root(self, link, pname, shape):
        """a tensor of a module link as a learnable root leaf of the program (its Parameter is a segment of the parameter buffer)"""
        key = (id(link), pname)
        hit = self.module_roots.get(key)
        if hit is None:
            hit = self.module_roots[key] = _ModuleTensor(link, pname)
        return self.mk("root", (), hit, shape)

    def root_node(self, var):
        """a RootVariable leaf; its rank inside links is that of its stored value minus the sample axis"""
        v = var.value if not var.learnable else var.parameter.numpy()
        return self.ranked(self.mk("root", (), var, self.root_shape(var)), min(max(np.ndim(v) - 1, 1), 3))

    def z_node(self, var):
        """a sampled posterior variable; its sample has the rank its parameters broadcast to (`slot_rank`)"""
        return self.ranked(self.mk("z", (), var, self.slots[var].shape), self.slot_rank.get(var, 3))

    def sample_rank(self, params):
        """rank (inside links) of a draw from parameters `params`: `broadcast_and_squeeze` (`utilities.py:143-149`) views
        them [N, B, 1, 1] when every one is a single element per row, else pads the shorter ones with ONE trailing axis
        (`uniform_shapes`, `utilities.py:274-279`) — the draw has the longest rank"""
        if all(p.shape[1] * p.shape[2] == 1 for p in params):
            return 3
        return max([self.rank_of(p) for p in params if p.op != "imm"] or [3])

    def check_ranks(self, args, what):
        """torch aligns TRAILING axes: [rows, d] op [rows, d1, d2] would pair the row axis with d1 (the reference then
        silently mixes Monte-Carlo samples).  Refused instead of reproduced."""
        ranks = {self.rank_of(a) for a in args if a.op != "imm"}
        if len(ranks) > 1:
            raise LoweringError("%s combines link values of different rank (an integer index or a sum without keepdim "
                                "dropped an axis on one side, or a 1-D array meets a matrix): torch would broadcast the "
                                "sample axis against an element axis.  Use keepdim=True / a slice / arrays of equal rank"
                                % what)
        return ranks.pop() if ranks else 3

    def view_node(self, op, arg, attr, shape, rank):
        node = self.mk(op, (arg,), attr, shape)
        if rank != 3:
            self.view_rank[node.key] = rank
        return node

    def view_call(self, fn, arg, rest, kwargs):
        kwargs = dict(kwargs)
        rank = self.rank_of(arg)
        B, D1, D2 = arg.shape

        def axis_of(x, what):
            d = self.static_int(x, what)
            d = d + rank if d < 0 else d
            if not 1 <= d < rank:
                raise LoweringError("%s=%d: axis 0 of a link value is the (sample, datapoint) axis and cannot be reduced or "
                                    "moved; element axes are 1..%d" % (what, d, rank - 1))
            return d

        if fn == "sum":
            dim = kwargs.pop("dim", kwargs.pop("axis", rest[0] if rest else None))
            keep = kwargs.pop("keepdim", kwargs.pop("keepdims", rest[1] if len(rest) > 1 else False))
            if dim is None or kwargs:
                raise LoweringError("BF.sum needs an explicit dim (and only dim / keepdim): a full reduction would sum over "
                                    "Monte-Carlo samples")
            d, keep = axis_of(dim, "BF.sum dim"), bool(self.static_int(keep, "BF.sum keepdim"))
            # both element axes reduced, and too many terms to unroll: the reduce node (reduce_external)
            if keep and rank == 3 and arg.op == "vsum" and arg.attr[1] and arg.attr[0] != d:
                inner = arg.args[0]
                if self.rank_of(inner) == 3 and _nelem(inner.shape) > self.kMaxViewTerms:
                    return self.reduce_external(inner, "BF.sum(BF.sum(...)) over %d x %d elements" % inner.shape[1:])
            if d == 1:
                shape = (B, 1, D2) if keep else (B, D2, 1)
            else:
                shape = (B, D1, 1)
            return self.view_node("vsum", arg, (d, keep), shape, rank if keep else rank - 1)
        d0 = axis_of(kwargs.pop("dim0", rest[0] if rest else None), "BF.transpose dim0")
        d1 = axis_of(kwargs.pop("dim1", rest[1] if len(rest) > 1 else None), "BF.transpose dim1")
        if kwargs:
            raise LoweringError("BF.transpose takes dim0 and dim1")
        if d0 == d1:
            return arg
        return self.view_node("vperm", arg, None, (B, D2, D1), rank)

    def view_index(self, arg, key):
        """`x[key]`: key = (whole sample axis, k1[, k2]) with integers (the axis is dropped) or plain slices"""
        if not isinstance(key, tuple) or not key or key[0] != slice(None, None, None) or len(key) > 3:
            raise LoweringError("unsupported index %r of a link value" % (key,))
        rank = self.rank_of(arg)
        extents = list(arg.shape[1:rank])
        if len(key) - 1 > len(extents):
            raise LoweringError("too many indices %r for a link value with %d element axes" % (key[1:], len(extents)))
        picks = []                                   # per element axis of the argument: ("int", i) | ("range", start, n)
        for ax, n in enumerate(extents):
            k = key[1 + ax] if 1 + ax < len(key) else slice(None, None, None)
            if isinstance(k, slice):
                if k.step not in (None, 1):
                    raise LoweringError("strided slices of link values are not lowered")
                start, stop, _ = k.indices(n)
                picks.append(("range", start, max(stop - start, 0)))
            else:
                i = self.static_int(k, "index")
                i = i + n if i < 0 else i
                if not 0 <= i < n:
                    raise IndexError("index %d is out of bounds for an element axis of extent %d" % (i, n))
                picks.append(("int", i))
        kept = [p[2] for p in picks if p[0] == "range"]
        if any(n == 0 for n in kept):
            raise LoweringError("empty slice of a link value")
        shape = (arg.shape[0],) + tuple(kept) + (1,) * (2 - len(kept))
        return self.view_node("vindex", arg, tuple(picks), shape, 1 + len(kept))

    def has_view(self, node):
        hit = self.view_memo.get(node.key)
        if hit is None:
            hit = node.op in ("vsum", "vperm", "vindex", "vstack") or any(self.has_view(a) for a in node.args)
            self.view_memo[node.key] = hit
        return hit

    def element_of(self, node, idx):
        """scalar IR for element idx = (b, i, j) of `node` (axes of extent 1 broadcast)"""
        idx = tuple(0 if node.shape[a] == 1 else idx[a] for a in range(3))
        key = (node.key, idx)
        hit = self.elem_memo.get(key)
        if hit is not None:
            return hit
        b, i, j = idx
        if node.op == "imm" or (node.shape == (1, 1, 1) and not self.has_view(node)):
            out = node
        elif node.op in ("z", "obs", "root", "carr") or (not self.has_view(node) and self.match_uniform(node) is not None):
            out = self.mk("elem", (node,), (b * node.shape[1] + i) * node.shape[2] + j, (1, 1, 1))
        elif node.op == "vsum":
            arg, (d, keep) = node.args[0], node.attr
            if d == 1:
                terms = [self.element_of(arg, (b, k, j if keep else i)) for k in range(arg.shape[1])]
            else:
                terms = [self.element_of(arg, (b, i, k)) for k in range(arg.shape[2])]
            self.view_terms += len(terms)
            if self.view_terms > self.kMaxViewTerms:
                raise LoweringError("BF.sum inside a link is unrolled per element: more than %d terms (use a matmul link, "
                                    "which runs on the dense path)" % self.kMaxViewTerms)
            out = terms[0]
            for t in terms[1:]:
                out = self.mk("add", (out, t))
        elif node.op == "vperm":
            out = self.element_of(node.args[0], (b, j, i))
        elif node.op == "vindex":
            src, free = [], [i, j]
            for p in node.attr:
                src.append(p[1] if p[0] == "int" else p[1] + free.pop(0))
            src += [0] * (2 - len(src))
            out = self.element_of(node.args[0], (b, src[0], src[1]))
        elif node.op == "vstack":
            out = node.args[j]
        elif node.op == "elem":
            out = node
        else:
            out = self.mk(node.op, tuple(self.element_of(a, idx) for a in node.args), node.attr)
        self.elem_memo[key] = out
        return out

    def split_elements(self, var, value, params, shape):
        """a model term whose value / parameters contain views -> one scalar term per element (like `mvn_terms`)"""
        nodes = ([value] if value is not None else []) + list(params)
        if not any(self.has_view(n) for n in nodes):
            return None

        class _Element:                            # what the emission reads of a model variable
            def __init__(self, i):
                self.is_observed, self.distribution, self.b_axis = var.is_observed, var.distribution, shape[0]
                self.name = var.name if i is None else "%s[%s]" % (var.name, ",".join(map(str, i)))

        out = []
        for idx in np.ndindex(*shape):
            single = shape == (1, 1, 1)
            out.append((_Element(None if single else idx), None if value is None else self.element_of(value, idx),
                        [self.element_of(p, idx) for p in params], (1, 1, 1)))
        return out

    # ---------------------------------------------------------------- model contexts
    def q_value(self, var):
        if isinstance(var, RootVariable):
            return self.root_node(var)
        if getattr(var, "_type", None) == "Deterministic node":
            return self.from_expr(var.link.expressions()["value"].expr, self.q_value)
        if isinstance(var, RandomVariable):
            if var not in self.slots:
                raise LoweringError("posterior variable %r is used before it is sampled" % var.name)
            return self.z_node(var)
        raise LoweringError("unsupported posterior variable %r" % (var,))

    def mean_value(self, var):
        """Taylor1Estimator (`gradient_estimators.py:47-56`): f is evaluated at
        ``means[v] = v._get_mean(input_values=samples)`` — the analytic mean of q_v given the SAMPLED values of its
        parents (`variables.py:85-86,522-525`, `distributions.py:77-82,126-139`) — for every variable of the sampler;
        everything else keeps its sampled value.  The samples still carry gradient (the ``differentiable=False`` of
        `gradient_estimators.py:50` is dropped at `variables.py:567`), so the means back-propagate into
        the parents' reparameterised draws.  Normal: mean = loc (torch normal.py:55-56)."""
        if isinstance(var, RootVariable) or getattr(var, "_type", None) == "Deterministic node":
            return self.q_value(var)          # the mean of a deterministic node is its value on the sampled parents
        if not isinstance(var, RandomVariable) or var not in self.slots:
            raise LoweringError("posterior variable %r is used before it is sampled" % getattr(var, "name", var))
        params = self.node_params(var, self.q_value)
        if var.distribution.kind == D.DIST_NORMAL:
            return params[0]
        if var.distribution.kind == D.DIST_LOGNORMAL:          # exp(loc + scale^2 / 2)   (torch log_normal.py:53-54)
            half_var = self.mk("mul", (self.mk("imm", (), 0.5), self.mk("call:square", (params[1],))))
            return self.mk("call:exp", (self.mk("add", (params[0], half_var)),))
        if var.distribution.kind == D.DIST_LAPLACE:            # loc   (torch laplace.py:51-52)
            return params[0]
        if var.distribution.kind == D.DIST_BETA:               # c1 / (c1 + c0)   (torch beta.py:63-64)
            return self.mk("truediv", (params[0], self.mk("add", (params[0], params[1]))))
        if var.distribution.kind == D.DIST_BERNOULLI:          # probs = sigmoid(logits)   (torch bernoulli.py:82-83)
            return self.mk("call:sigmoid", (params[0],))
        if var.distribution.kind == D.DIST_BINOMIAL:           # total_count * probs   (torch binomial.py:98-99)
            return self.mk("mul", (params[0], self.mk("call:sigmoid", (params[1],))))
        raise LoweringError("Taylor1 estimator: %r (%s) has no analytic mean (torch returns NaN for a Cauchy)"
                            % (var.name, type(var.distribution).__name__))

    def p_value(self, var):
        if isinstance(var, RandomVariable) and var.is_observed:
            if (not var.has_observed_value and getattr(var, "_type", None) == "Deterministic node"
                    and not getattr(var, "has_random_dataset", False)):
                # DeterministicVariable(data, name, is_observed=True) — the regressors of
                # examples/multivariate_regression.py: the value is the link of its own observed root
                # (datapoint axis first), `standard_variables.py:37-68,115-130`
                return self.from_expr(var.link.expressions()["value"].expr, self.p_value)
            if not var.has_observed_value:
                # the minibatch data path on the scalar engine (`standard_variables.py:71-112`,
                # `distributions.py:393-473`): the value is `batch_size` rows of a dataset, other rows in every evaluation — an
                # observed EmpiricalVariable itself, or a variable observed THROUGH one (`variables.py:572-590`).  To the program these
                # are ordinary observations [B, ...] in the observation buffer; the engine refreshes them in front of every launch
                # (`bsvi_minibatch_gather`: the keyed bijection of the dense path's `dense_head`, or the caller's rows).
                source = var if getattr(var, "_type", None) == "Empirical" else getattr(var, "dataset", None)
                if source is None and getattr(var, "_type", None) != "Empirical":
                    # observed BY FLAG only (`NormalVariable(..., is_observed=True)` never given a value — the stimulus node of
                    # examples/PopulationReceptiveFields.py:29): the reference DRAWS it from its own distribution once per evaluation
                    # (`variables.py:849` takes observed_submodel._get_sample(1, observed=True), `:553-565` draws what has no value)
                    handle = self.drawn_handle(var)
                    return self.mk("obs", (), handle, canonical_elem_shape(handle._observed_value.shape[1:]))
                if source is None or getattr(source, "_type", None) != "Empirical":
                    raise LoweringError("variable %r is observed without a value and not through an EmpiricalVariable" % var.name)
                handle = self.minibatch_handle(source)
                return self.mk("obs", (), handle, canonical_elem_shape(handle._observed_value.shape[1:]))
            return self.mk("obs", (), var, canonical_elem_shape(var._observed_value.shape[1:]))
        if var.name in self.q_by_name:
            qv = self.q_by_name[var.name]
            if self.estimator == "taylor1":
                return self.mean_value(qv)
            return self.q_value(qv)
        if isinstance(var, RootVariable):
            return self.root_node(var)
        if getattr(var, "_type", None) == "Deterministic node":
            return self.from_expr(var.link.expressions()["value"].expr, self.p_value)
        raise LoweringError("model variable %r is neither observed nor present in the posterior "
                            "(the reference raises AttributeError here, variables.py:430)" % var.name)

    # ---------------------------------------------------------------- tables
    def param_offset(self, param, group):
        off = self.param_index.get(id(param))
        if off is None:
            off = self.n_params
            self.param_index[id(param)] = off
            self.parameters.append((param, off, param.size, group))
            self.n_params += param.size
        return off

    def const_offset(self, arr, key):
        off = self.const_index.get(key)
        if off is None:
            off = self.n_consts
            self.const_index[key] = off
            flat = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
            self.consts.append(flat)
            self.n_consts += flat.size
        return off

    def host_value(self, node):
        """numpy value [B, D1, D2] (broadcastable) of an IR node made of constants, non-learnable roots and OBSERVED values only
        (else None): the parameters of a node that is drawn once per evaluation are constants of the model"""
        if node.has_z:
            return None
        if node.op == "imm":
            return np.full((1, 1, 1), float(node.attr))
        if node.op == "carr":
            return np.asarray(node.attr, dtype=np.float64).reshape(node.shape)
        if node.op == "root":
            if node.attr.learnable:
                return None
            return np.asarray(node.attr.value, dtype=np.float64).reshape(node.shape)
        if node.op == "obs":
            if isinstance(node.attr, (MinibatchObs, DrawnObs)):
                return None
            return np.asarray(node.attr._observed_value, dtype=np.float64).reshape(node.shape)
        args = [self.host_value(a) for a in node.args]
        if any(a is None for a in args):
            return None
        if node.op in ("add", "sub", "mul", "truediv", "pow"):
            fn = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "truediv": np.divide, "pow": np.power}[node.op]
            return fn(args[0], args[1])
        if node.op.startswith("call:"):
            g = node.op[5:]
            table = dict(exp=np.exp, log=np.log, sqrt=np.sqrt, sin=np.sin, cos=np.cos, tanh=np.tanh, abs=np.abs, square=np.square,
                         neg=np.negative, sigmoid=lambda x: 1.0 / (1.0 + np.exp(-x)), softplus=lambda x: np.logaddexp(0.0, x))
            if g in table and len(args) == 1:
                return table[g](args[0])
        return None

    def drawn_handle(self, var):
        """the observation record behind a variable that is observed by flag only: a Normal whose parameters are constants of the model,
        drawn once per evaluation and shared by all samples"""
        hit = self.drawn_obs.get(id(var))
        if hit is not None:
            return hit
        if var.distribution.kind != D.DIST_NORMAL:
            raise LoweringError("variable %r is observed without a value: only a Normal node is drawn per evaluation here" % var.name)
        loc, scale = self.node_params(var, self.p_value)
        mean, sd = self.host_value(loc), self.host_value(scale)
        if mean is None or sd is None:
            raise LoweringError("variable %r is observed without a value and its parameters are not constants of the model" % var.name)
        shape = broadcast_shapes3(loc.shape, scale.shape)
        handle = DrawnObs(var, np.broadcast_to(mean, shape).astype(np.float32), np.broadcast_to(sd, shape).astype(np.float32))
        self.drawn_obs[id(var)] = handle
        return handle

    def reduce_external(self, prod, what):
        """A link reduces `prod` [B, D1, D2] over BOTH element axes and the product is too large to unroll (B x D1 x D2 terms beyond
        kMaxViewTerms): `BF.sum(BF.sum(receptive_field * input, dim=1), dim=2)` of examples/PopulationReceptiveFields.py:29-31.  With
        prod = A * X — A [1, D1, D2] an ELEMENTWISE expression of constant matrices and scalars that are sampled or learnable, X
        [B, D1, D2] observed data (a value, or a node DRAWN once per evaluation: `drawn_handle`) — the reduction leaves the per-sample
        program for the library's reduce node (`bsvi_reduce_*`, csrc/reduce_kernel.h): r_d = sum_e A(e; s) X[d][e] re-enters as
        r_d = e_d + sum_k g_dk s_k, value and gradient exact at the sample, composed here from GIVEN rows the node fills (K + 1 pseudo
        posterior variables of B elements each, behind the posterior's own rows — the mechanism of `mvn_external`)."""
        if prod.op != "mul":
            raise LoweringError("%s: a reduction over %d elements is lowered for a product  expression * data  only" % (what, _nelem(prod.shape)))
        a, x = prod.args
        if x.op != "obs":
            a, x = x, a
        B, D1, D2 = prod.shape
        if x.op != "obs" or x.shape != (B, D1, D2) or a.shape != (1, D1, D2):
            raise LoweringError("%s: the reduced product must be  expression [1, %d, %d] * observed data [%d, %d, %d]" % (what, D1, D2, B, D1, D2))
        if isinstance(x.attr, MinibatchObs):
            raise LoweringError("%s: a minibatch as the data of a large reduction is not lowered yet" % what)
        if self.pseudo_q:
            raise LoweringError("%s: a second external node (its rows would lie behind the first one's)" % what)
        host_g = dict(identity=lambda v: v, softplus=lambda v: np.logaddexp(0.0, v), sigmoid=lambda v: 1.0 / (1.0 + np.exp(-v)),
                      exp=np.exp, log=np.log, tanh=np.tanh, sqrt=np.sqrt, square=np.square)
        code, mats, memo = [], [], {}
        slot_inputs, uniform_inputs = [], []

        def push(ins):
            code.append(ins)
            return len(code) - 1

        def emit(node):
            hit = memo.get(node.key)
            if hit is not None:
                return hit
            if node.shape[0] != 1 or any(n not in (1, d) for n, d in zip(node.shape[1:], (D1, D2))):
                raise LoweringError("%s: the expression under the reduction mixes shapes (%r over a %dx%d field)" % (what, node.shape, D1, D2))
            arr = self.host_value(node)
            if arr is not None:
                arr = np.broadcast_to(arr, (1, D1, D2))[0]
                if np.all(arr == arr.flat[0]):
                    t = push(("IMM", 0, 0, 0, float(arr.flat[0])))
                else:
                    mats.append(np.ascontiguousarray(arr, dtype=np.float32))
                    t = push(("MAT", 0, len(mats) - 1, 0, 0.0))
            elif node.op == "z" or (node.op == "elem" and node.args[0].op == "z"):
                if node.shape != (1, 1, 1):
                    raise LoweringError("%s: a sampled VECTOR under the reduction (only scalars per sample are inputs of the reduce node)" % what)
                child, j = (node, 0) if node.op == "z" else (node.args[0], int(node.attr))
                slot_inputs.append((node, self.slots[child.attr].base + j))
                t = push(("INPUT", 0, ("s", len(slot_inputs) - 1), 0, 0.0))
            elif not node.has_z and self.match_uniform(node) is not None:
                leaf, g, aa, bb = self.match_uniform(node)
                if _nelem(leaf.shape) != 1:
                    raise LoweringError("%s: a learnable ARRAY under the reduction (only scalars are inputs of the reduce node)" % what)
                is_param, k0 = self.uniform_entries(leaf, g, aa, bb)
                uniform_inputs.append((node, self.uni_param[k0]))
                t = push(("INPUT", 0, ("u", len(uniform_inputs) - 1), 0, 0.0))
            elif node.op == "pow" and node.args[1].op == "imm":
                t = push(("UN", UNOP["powi"], emit(node.args[0]), 0, float(node.args[1].attr)))
            elif node.op in BINOP and node.op != "delta":
                xx, yy = emit(node.args[0]), emit(node.args[1])
                t = push(("BIN", BINOP[node.op], xx, yy, 0.0))
            elif node.op.startswith("call:") and node.op[5:] in UNOP and node.op[5:] not in ("p2l", "relu", "log1p", "expm1"):
                t = push(("UN", UNOP[node.op[5:]], emit(node.args[0]), 0, 0.0))
            else:
                raise LoweringError("%s: %s under the reduction is not served by the reduce node" % (what, node.op))
            memo[node.key] = t
            return t

        emit(a)
        K = len(slot_inputs) + len(uniform_inputs)
        if K > 8:
            raise LoweringError("%s: more than 8 sampled / learnable scalars under the reduction" % what)
        n_s = len(slot_inputs)
        code = [(k, f, (aa[1] if aa[0] == "s" else n_s + aa[1]) if isinstance(aa, tuple) else aa, bb, imm) for k, f, aa, bb, imm in code]
        node = ExternalReduce()
        node.name, node.rows, node.cols, node.n_data, node.code = what, D1, D2, B, code
        node.mats = np.stack(mats) if mats else np.zeros((0, D1, D2), np.float32)
        node.slot_inputs = [row for _, row in slot_inputs]
        node.uniform_inputs = np.zeros(len(uniform_inputs), dtype=UNIFORM_DTYPE)
        for k, (_, (src, tr, is_param, aa, bb)) in enumerate(uniform_inputs):
            node.uniform_inputs[k] = (src, tr, is_param, 0, aa, bb)
        handle = x.attr
        node.drawn = isinstance(handle, DrawnObs)
        if node.drawn:
            node.data_mean, node.data_scale = handle.mean.reshape(B, D1 * D2), handle.scale.reshape(B, D1 * D2)
            node.drawn_name, node.weight = handle.name, 1.0
            handle.consumed = True
        else:
            node.data_mean, node.data_scale = np.asarray(handle._observed_value, dtype=np.float32).reshape(B, D1 * D2), None
            node.drawn_name, node.weight = None, 0.0
        node.n_rows_out = (K + 1) * B + 1
        self.externals.append(node)
        partners = [n for n, _ in slot_inputs] + [n for n, _ in uniform_inputs]
        if self.external_mode == "omit":
            # the base program makes the draw only: the reduction's value does not matter there (the model term that reads it is
            # evaluated, its result unused) — a constant keeps the record well formed
            return self.mk("carr", (), np.zeros((B, 1, 1), dtype=np.float32), (B, 1, 1))
        node.row0 = self.n_slots

        class _Rows:                                       # a pseudo posterior variable of B elements: its "noise" rows are GIVEN
            _pseudo, is_observed = True, False

            def __init__(self, name):
                self.name, self.distribution = name, D.NormalDistribution()

        def given(name, shape):
            rows = _Rows(name)
            self.slots[rows] = SlotInfo(rows, self.n_slots, shape, D.DIST_NORMAL)
            self.n_slots += _nelem(shape)
            self.pseudo_q.append(rows)
            return self.mk("z", (), rows, shape)

        out = None
        for k in range(K):
            term = self.mk("mul", (given("%s/gradient[%d]" % (what, k), (B, 1, 1)), partners[k]))
            out = term if out is None else self.mk("add", (out, term))
        e = given("%s/value" % what, (B, 1, 1))
        out = e if out is None else self.mk("add", (e, out))
        node.logp_node = given("%s/drawn log-probability" % what, (1, 1, 1))
        return out

    def minibatch_handle(self, source):
        """the observation record behind an observed EmpiricalVariable: B rows of its dataset, refreshed per evaluation"""
        hit = self.minibatch_obs.get(id(source))
        if hit is not None:
            return hit
        from brancher_amd.standard_variables import RandomIndices
        if not source.is_observed:
            raise LoweringError("the EmpiricalVariable %r must be observed (is_observed=True) to feed the joint model" % source.name)
        exprs = source.link.expressions()
        ds = exprs["dataset"].expr
        if ds.op != "var" or not isinstance(ds.attr, RootVariable) or "weights" in exprs:
            raise LoweringError("the EmpiricalVariable %r must hold an array dataset (no weights)" % source.name)
        data = np.asarray(ds.attr.value, dtype=np.float32)          # observed datasets are stored [1, DS, ...] (utilities.py:226-232)
        if data.ndim < 2 or data.shape[0] != 1:
            raise LoweringError("the dataset of %r has an unexpected layout %r" % (source.name, data.shape))
        data = data[0]
        indices = None
        if "indices" in exprs:
            ind = exprs["indices"].expr
            if ind.op != "var" or not isinstance(ind.attr, RandomIndices):
                raise LoweringError("the EmpiricalVariable %r must be indexed by a RandomIndices variable (or draw its own batch_size rows)" % source.name)
            indices = ind.attr
            if int(np.asarray(indices.link.expressions()["dataset"].expr.attr.value).size) != data.shape[0]:
                raise LoweringError("the RandomIndices variable of %r ranges over another dataset size" % source.name)
        handle = MinibatchObs(source, data, int(source.batch_size), indices)
        self.minibatch_obs[id(source)] = handle
        return handle

    def obs_offset(self, var):
        hit = self.obs_index.get(id(var))
        if hit is None:
            flat = np.ascontiguousarray(var._observed_value, dtype=np.float32).reshape(-1)
            hit = self.n_obs
            self.obs_index[id(var)] = hit
            self.obs.append(flat)
            self.n_obs += flat.size
        return hit

    def uniform_entries(self, leaf, transform, a, b):
        """allocate (or reuse) U entries a + b*g(leaf) for every element of a root/const leaf."""
        ukey = (leaf.key, transform, float(a), float(b))
        hit = self.uni_index.get(ukey)
        if hit is not None:
            return hit
        size = _nelem(leaf.shape)
        if leaf.op == "root" and leaf.attr.learnable:
            src0 = self.param_offset(leaf.attr.parameter, self.group_of(leaf.attr))
            table, is_param = self.uni_param, 1
        elif leaf.op == "root":
            src0 = self.const_offset(leaf.attr.value, ("root", id(leaf.attr)))
            table, is_param = self.uni_const, 0
        else:  # carr
            src0 = self.const_offset(leaf.attr, leaf.key)
            table, is_param = self.uni_const, 0
        k0 = len(table)
        for e in range(size):
            table.append((src0 + e, UT[transform], is_param, a, b))
        self.uni_index[ukey] = (is_param, k0)
        return is_param, k0

    def group_of(self, root):
        return 0 if root in self.q_roots else 1

    # ---------------------------------------------------------------- uniform pattern matcher
    def match_uniform(self, node):
        """node == a + b*g(leaf) with leaf a root/array constant?  -> (leaf, g, a, b) or None"""
        if node.has_z:
            return None
        if node.op in ("root", "carr"):
            return node, "identity", 0.0, 1.0
        if node.op.startswith("call:"):
            g = node.op[5:]
            if g in UT and node.args[0].op in ("root", "carr"):
                return node.args[0], g, 0.0, 1.0
            return None
        if node.op in ("add", "sub", "mul", "truediv"):
            x, y = node.args
            if x.op == "imm" and y.op != "imm":
                m = self.match_uniform(y)
                if m is None:
                    return None
                leaf, g, a, b = m
                c = x.attr
                if node.op == "add":
                    return leaf, g, c + a, b
                if node.op == "sub":
                    return leaf, g, c - a, -b
                if node.op == "mul":
                    return leaf, g, c * a, c * b
                return None
            if y.op == "imm" and x.op != "imm":
                m = self.match_uniform(x)
                if m is None:
                    return None
                leaf, g, a, b = m
                c = y.attr
                if node.op == "add":
                    return leaf, g, a + c, b
                if node.op == "sub":
                    return leaf, g, a - c, b
                if node.op == "mul":
                    return leaf, g, a * c, b * c
                if node.op == "truediv":
                    return leaf, g, a / c, b / c
        return None

    # ---------------------------------------------------------------- code generation
    def is_leaf_operand(self, node):
        return node.op in ("z", "obs", "imm", "elem") or self.match_uniform(node) is not None

    def count_uses(self, roots):
        """per record: every computed (non-leaf, sample-dependent) sub-expression counts once"""
        seen = set()
        stack = list(roots)
        while stack:
            n = stack.pop()
            if n.key in seen or self.is_leaf_operand(n):
                continue
            seen.add(n.key)
            stack.extend(n.args)
        for k in seen:
            self.use_count[k] = self.use_count.get(k, 0) + 1

    def begin_record(self, shape, sink=False, k=0, outer=()):
        """One record covers the elements of `shape` whose first k indices are `outer` (k = 0: all of them)."""
        self.rec_sink = sink
        self.rec_shape = tuple(shape)
        self.rec_k, self.rec_outer = k, tuple(outer)
        self.rec_elems = _nelem(self.rec_shape[k:])
        self.rec_begin = len(self.code)
        self.rec_operands = {}
        self.rec_ntemp = 0

    def end_record(self):
        self.records.append((self.rec_begin, len(self.code), self.rec_elems, self.rec_ntemp, self.rec_sink))
        self.max_temps = max(self.max_temps, self.rec_ntemp)

    def place(self, leaf_shape):
        """(element offset inside the leaf, stride flag) of a leaf operand in the current record."""
        s = tuple(leaf_shape)
        if _nelem(s) == 1:
            return 0, 0
        for i in range(3):
            if s[i] not in (1, self.rec_shape[i]):
                raise LoweringError("a %r operand cannot be broadcast inside a %r node" % (s, self.rec_shape))
        k = self.rec_k
        inner = [i for i in range(k, 3) if self.rec_shape[i] > 1]
        full = all(s[i] == self.rec_shape[i] for i in inner)
        flat = all(s[i] == 1 for i in inner)
        if not (full or flat):
            raise LoweringError("partial broadcasting of a %r operand inside a %r node is not supported by the "
                                "fused kernel" % (s, self.rec_shape))
        strides = (s[1] * s[2], s[2], 1)
        offset = sum((self.rec_outer[i] if s[i] > 1 else 0) * strides[i] for i in range(k))
        return offset, (1 if (full and inner and any(s[i] > 1 for i in inner)) else 0)

    def leaf_shapes(self, roots):
        """shapes of every leaf operand (and shared derived value) the expressions in `roots` read"""
        out, seen, stack = [], set(), [r for r in roots if r is not None]
        while stack:
            n = stack.pop()
            if n.key in seen:
                continue
            seen.add(n.key)
            m = self.match_uniform(n) if n.op != "elem" else None
            if n.op == "elem":
                out.append((1, 1, 1))
            elif m is not None:
                out.append(tuple(m[0].shape))
            elif n.op in ("z", "obs", "root", "carr") or n.key in self.derived_nodes:
                out.append(tuple(n.shape))
            else:
                stack.extend(n.args)
        return out

    def for_each_record(self, shape, roots, sink, body):
        """emit `body()` once per record of a node of `shape` reading the expressions `roots` (see _split_axis)"""
        k = _split_axis(shape, self.leaf_shapes(roots) + [tuple(shape)])
        for outer in (np.ndindex(*shape[:k]) if k else [()]):
            self.begin_record(shape, sink=sink, k=k, outer=outer)
            body()
            self.end_record()

    def put(self, op, flags=0, dist=0, dst=None, a=None, b=None, c=None, s=None, imm0=0.0, imm1=0.0):
        w0 = OP[op] | (flags << 8) | (dist << 16)
        none = operand(K_NONE)
        self.code.append([w0, dst or none, a or none, b or none, c or none, s or none, _fbits(imm0), _fbits(imm1)])

    def const_operand(self, value):
        """a literal constant lives in the constant buffer and is read through the uniform table"""
        key = ("imm", float(np.float32(value)))
        off = self.const_offset(np.array([value], dtype=np.float32), key)
        ukey = (key, "identity", 0.0, 1.0)
        hit = self.uni_index.get(ukey)
        if hit is None:
            self.uni_const.append((off, UT["identity"], 0, 0.0, 1.0))
            hit = (0, len(self.uni_const) - 1)
            self.uni_index[ukey] = hit
        return hit

    def operand_of(self, node):
        """operand for an IR node; computed sub-expressions are materialised into a temp (or a
        shared derived slot) by instructions emitted in front of the consumer"""
        hit = self.rec_operands.get(node.key)
        if hit is not None:
            return hit
        m = self.match_uniform(node) if node.op != "elem" else None
        if node.op == "elem":
            # element `attr` (flattened index) of a vector-valued leaf: a fixed operand, no element loop
            child, j = node.args[0], int(node.attr)
            mc = self.match_uniform(child)
            if child.op == "z":
                res = operand(K_Z, self.slots[child.attr].base + j, 0)
            elif child.op == "obs":
                res = operand(K_OBS, self.obs_offset(child.attr) + j, 0)
            elif mc is not None:
                leaf, g, a, b = mc
                is_param, k0 = self.uniform_entries(leaf, g, a, b)
                res = operand(K_U if is_param else K_UCONST, k0 + j, 0)
            else:
                raise LoweringError("an element of a computed vector (%r) is not addressable in the fused kernel" % (child,))
        elif m is not None:
            leaf, g, a, b = m
            is_param, k0 = self.uniform_entries(leaf, g, a, b)
            off, stride = self.place(leaf.shape)
            res = operand(K_U if is_param else K_UCONST, k0 + off, stride)
        elif node.op == "imm":
            _, k0 = self.const_operand(node.attr)
            res = operand(K_UCONST, k0, 0)
        elif node.op == "z":
            slot = self.slots[node.attr]
            off, stride = self.place(slot.shape)
            res = operand(K_Z, slot.base + off, stride)
        elif node.op == "obs":
            off, stride = self.place(node.shape)
            res = operand(K_OBS, self.obs_offset(node.attr) + off, stride)
        elif node.key in self.derived:
            off, stride = self.place(node.shape)
            res = operand(K_Z, self.derived[node.key] + off, stride)
        else:
            t = self.rec_ntemp
            self.rec_ntemp += 1
            res = operand(K_Z, self.temp_base + t, 0)
            self.emit_compute(node, res)
        self.rec_operands[node.key] = res
        return res

    def emit_compute(self, node, dst):
        if node.op == "pow" and node.args[1].op == "imm":
            self.put("UN", flags=UNOP["powi"], dst=dst, a=self.operand_of(node.args[0]), imm0=node.args[1].attr)
        elif node.op in BINOP:
            a, b = self.operand_of(node.args[0]), self.operand_of(node.args[1])
            self.put("BIN", flags=BINOP[node.op], dst=dst, a=a, b=b)
        elif node.op.startswith("call:"):
            self.put("UN", flags=UNOP[node.op[5:]], dst=dst, a=self.operand_of(node.args[0]))
        else:
            raise LoweringError("cannot generate code for %r" % (node,))

    def ensure_derived(self, roots):
        """emit the defining records of shared sub-expressions these roots need (before their first use)"""
        order = []
        seen = set()

        def visit(n):
            if n.key in seen or self.is_leaf_operand(n):
                return
            seen.add(n.key)
            for a in n.args:
                visit(a)
            if n.key in self.derived_nodes and n.key not in self.derived:
                order.append(n)

        for r in roots:
            visit(r)
        for n in order:
            base = self.n_latent + self.n_derived
            size = _nelem(n.shape)
            self.n_derived += size
            def body(n=n, base=base):
                off, stride = self.place(n.shape)
                self.emit_compute(n, operand(K_Z, base + off, stride))
            self.for_each_record(n.shape, list(n.args), False, body)
            self.derived[n.key] = base

    @staticmethod
    def affine_parts(loc, is_leaf):
        """loc == A*B + C with A, B, C arbitrary sub-expressions (None = neutral element)"""
        if is_leaf(loc):
            return loc, None, None
        if loc.op == "add":
            x, y = loc.args
            if x.op == "mul" and not is_leaf(x):
                return x.args[0], x.args[1], y
            if y.op == "mul" and not is_leaf(y):
                return y.args[0], y.args[1], x
            return x, None, y
        if loc.op == "mul":
            return loc.args[0], loc.args[1], None
        return loc, None, None

    def emit_node(self, dist, flags, params, value=None, slot=None, w_lp=0.0, w_ent=0.0):
        """one node instruction (NAFF for Normal, NODE otherwise) plus whatever temps its operands need"""

        def opnd(node, neutral=0.0):
            # an absent factor/addend reads the constant 1.0 / 0.0 from the uniform table, so the
            # kernel loads every operand with the same branch-free ds_read
            if node is None:
                return operand(K_UCONST, self.const_operand(neutral)[1], 0)
            return self.operand_of(node)

        if flags & F_SAMPLE:
            off, stride = self.place(slot.shape)
            dst = operand(K_Z, slot.base + off, stride)
        else:
            dst = opnd(value)
        if dist == D.DIST_NORMAL:
            A, B, C = self.affine_parts(params[0], self.is_leaf_operand)
            a, b, c, s = opnd(A, 1.0), opnd(B, 1.0), opnd(C, 0.0), opnd(params[1], 1.0)
            self.put("NAFF", flags=flags, dist=dist, dst=dst, a=a, b=b, c=c, s=s, imm0=w_lp, imm1=w_ent)
        else:
            a = opnd(params[0])
            b = opnd(params[1]) if len(params) > 1 else opnd(None, 0.0)
            self.put("NODE", flags=flags, dist=dist, dst=dst, a=a, b=b, imm0=w_lp, imm1=w_ent)

    # ---------------------------------------------------------------- node parameter IR
    def node_params(self, var, ctx):
        links = var.link.expressions()
        var.distribution.check_parameters(**links)
        out = []
        for name, transform in var.distribution.resolve_kernel_parameters(links):
            node = self.from_expr(links[name].expr, ctx)
            if transform == "probs_to_logits":
                node = self.mk("call:p2l", (node,))
            out.append(node)
        return out

    # ---------------------------------------------------------------- multivariate normal terms
    def constant_value(self, e):
        """numpy value of a link expression made of constants and non-learnable roots only (else None)"""
        if e.op == "const":
            return np.asarray(e.attr, dtype=np.float64)
        if e.op == "var":
            var = e.attr
            if isinstance(var, RootVariable) and not var.learnable:
                return np.asarray(var.value, dtype=np.float64)
            return None
        args = [self.constant_value(a) if isinstance(a, sym.Expr) else np.asarray(a, dtype=np.float64) for a in e.args]
        if any(a is None for a in args):
            return None
        if e.op in sym.BINARY_OPS:
            fn = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "truediv": np.divide, "pow": np.power}[e.op]
            return fn(args[0], args[1])
        if e.op == "call" and isinstance(e.attr[0], str):
            name, kwargs = e.attr
            if name == "matmul" and len(args) == 2:
                return np.matmul(args[0], args[1])
            if name == "transpose":
                return np.swapaxes(args[0], int(args[1]), int(args[2])) if len(args) == 3 else np.swapaxes(args[0], -2, -1)
            if name in ("exp", "log", "sqrt", "abs", "sin", "cos", "tanh") and len(args) == 1 and not kwargs:
                return getattr(np, name)(args[0])
        return None

    def mvn_terms(self, v):
        """A MultivariateNormalVariable of the joint model (`standard_variables.py:317-347`, `distributions.py:314-331`)
        whose covariance is a constant of the model — the prior of a Gaussian process at fixed inputs
        (`stochastic_processes.py:29-40`, development_playgrounds/GP_playground.py).  With the Cholesky factor L of the
        covariance,  log N(x | m, L L^T) = sum_i log Normal(u_i | 0, L_ii)  for  u = diag(L) L^-1 (x - m):  D scalar
        Normal terms whose values are fixed linear combinations of the elements of x - m.  The factorisation is done here,
        once, in double precision; the kernel sees D Normal records (their reverse mode included).
        Returns [(term, value IR, [loc IR, scale IR], shape)] like the entries of p_nodes."""
        links = v.link.expressions()
        given = [k for k in ("scale_tril", "covariance_matrix", "precision_matrix") if k in links]
        mat = self.constant_value(links[given[0]].expr)
        if mat is None:
            return self.mvn_terms_symbolic(v, given[0])
        mat = np.asarray(mat, dtype=np.float64)
        mat = mat.reshape(mat.shape[-2:])
        if mat.shape[0] != mat.shape[1]:
            raise LoweringError("%s of %r is not a square matrix" % (given[0], v.name))
        if given[0] == "scale_tril":
            L = np.tril(mat)
        elif given[0] == "covariance_matrix":
            L = np.linalg.cholesky(mat)
        else:
            L = np.linalg.cholesky(np.linalg.inv(mat))
        dim = L.shape[0]
        A = np.diag(np.diag(L)) @ np.linalg.inv(L)
        value = self.p_value(v)
        loc = self.from_expr(links["loc"].expr, self.p_value)
        for what, node in (("value", value), ("loc", loc)):
            if _nelem(node.shape) not in (1, dim):
                raise LoweringError("the %s of %r has %d elements, its covariance is %dx%d" % (what, v.name, _nelem(node.shape), dim, dim))

        def element(node, j):
            if _nelem(node.shape) == 1:
                return node
            return self.mk("elem", (node,), j, (1, 1, 1))

        class _Term:                              # what the emission reads of a model variable
            def __init__(self, var, i):
                self.is_observed, self.name = var.is_observed, "%s[%d]" % (var.name, i)
                self.distribution = D.NormalDistribution()

        zero_loc = loc.op == "imm" and loc.attr == 0.0
        terms = []
        for i in range(dim):
            u = None
            for j in range(i + 1):
                d = element(value, j) if zero_loc else self.mk("sub", (element(value, j), element(loc, j)))
                t = self.mk("mul", (self.mk("imm", (), float(A[i, j])), d))
                u = t if u is None else self.mk("add", (u, t))
            terms.append((_Term(v, i), u, [self.mk("imm", (), 0.0), self.mk("imm", (), float(L[i, i]))], (1, 1, 1)))
        return terms

    kMaxSymbolicMvn = 10

    def mvn_terms_symbolic(self, v, given):
        """The same D Normal terms when the covariance depends on learnable or sampled values (a Gaussian process whose
        kernel hyper-parameters are inferred: `covariance_matrix = exp(-sqdist / (2 ell^2)) * amp + jitter` with a latent
        `ell`).  The Cholesky factorisation itself becomes part of the per-sample program, unrolled symbolically over the
        elements of the covariance link (Cholesky-Banachiewicz, D <= kMaxSymbolicMvn):
            L_ij = (c_ij - sum_{k<j} L_ik L_jk) / L_jj,   L_ii = sqrt(c_ii - sum_{k<i} L_ik^2),
        then forward substitution  u_i = d_i - sum_{j<i} L_ij w_j,  w_i = u_i / L_ii  with d = x - m, and
            log N(x | m, L L^T) = sum_i log Normal(u_i | 0, L_ii).
        ~D^3 / 3 multiply-adds of ordinary link arithmetic: the reverse mode through the factorisation comes for free, shared
        entries of L become derived slots (computed once per sample)."""
        links = v.link.expressions()
        mat = self.from_expr(links[given].expr, self.p_value)
        B, dim, dim2 = mat.shape
        if B != 1 or dim != dim2:
            raise LoweringError("%s of %r must be one square matrix per sample (shape %r)" % (given, v.name, mat.shape))
        if dim > self.kMaxSymbolicMvn or given == "precision_matrix":
            # the batched kernel serves all three parameterisations (bsvi_mvn_form): a scale_tril needs no factorisation there,
            # a precision matrix is factorised in place of the covariance (its inverse is never unrolled symbolically)
            return self.mvn_external(v, mat, given)
        value = self.p_value(v)
        loc = self.from_expr(links["loc"].expr, self.p_value)
        for what, node in (("value", value), ("loc", loc)):
            if _nelem(node.shape) not in (1, dim):
                raise LoweringError("the %s of %r has %d elements, its covariance is %dx%d" % (what, v.name, _nelem(node.shape), dim, dim))
        mk = self.mk

        def vec(node, j):                       # element j of a D-vector stored along whichever axis (or a scalar)
            if _nelem(node.shape) == 1:
                return self.element_of(node, (0, 0, 0))
            idx = [0, 0, 0]
            idx[[a for a in range(3) if node.shape[a] == dim][0]] = j
            return self.element_of(node, tuple(idx))

        L = [[None] * dim for _ in range(dim)]
        for i in range(dim):
            for j in range(i + 1):
                if given == "scale_tril":
                    L[i][j] = self.element_of(mat, (0, i, j))
                    continue
                acc = self.element_of(mat, (0, i, j))
                for k in range(j):
                    acc = mk("sub", (acc, mk("mul", (L[i][k], L[j][k]))))
                L[i][j] = mk("call:sqrt", (acc,)) if i == j else mk("truediv", (acc, L[j][j]))

        class _Term:
            def __init__(self, var, i):
                self.is_observed, self.name = var.is_observed, "%s[%d]" % (var.name, i)
                self.distribution = D.NormalDistribution()
                self.b_axis = 1

        zero_loc = loc.op == "imm" and loc.attr == 0.0
        terms, w = [], []
        for i in range(dim):
            u = vec(value, i) if zero_loc else mk("sub", (vec(value, i), vec(loc, i)))
            for j in range(i):
                u = mk("sub", (u, mk("mul", (L[i][j], w[j]))))
            w.append(mk("truediv", (u, L[i][i])))
            terms.append((_Term(v, i), u, [mk("imm", (), 0.0), L[i][i]], (1, 1, 1)))
        return terms

    kMaxExternalMvn = 1024        # (up to 192 the kernel keeps a sample's matrix in LDS, beyond it in a block of device memory)

    def mvn_external(self, v, mat, given="covariance_matrix"):
        """A MultivariateNormal term too large to unroll (D > kMaxSymbolicMvn) whose covariance is an ELEMENTWISE expression of
        constant matrices and SCALARS that are sampled or learnable: it leaves the per-sample program for the batched kernel
        of the library (`bsvi_mvn_*`, csrc/mvn_kernel.h: one wave per sample factorises the covariance in LDS).  Here the
        covariance link becomes three-address code over (matrix element, scalar input, immediate) leaves, and the term
        re-enters the program as a LINEAR surrogate  e + sum_k g_k * input_k  (BSVI_DIST_LINEAR records) whose per-sample
        coefficients are GIVEN rows the kernel fills — value and gradient of log p at the sample, so the program's reverse
        sweep carries d log p / d inputs on to the posterior's parameters.  (`distributions.py:314-331`,
        `standard_variables.py:317-347`.)"""
        # (the importance program — log p and log q at caller-supplied values, variables.py:821-841 — is served: its base
        #  program reports the supplied values as the "draw", the kernel evaluates the term there.  Taylor1 is not: it reads
        #  the model at the posterior's MEANS, which no program reports row by row)
        if self.estimator == "taylor1" and os.environ.get("BSVI_TAYLOR1_MVN", "1") == "0":
            raise LoweringError("%r: taylor1 through the batched multivariate-normal kernel is switched off" % v.name)
        links = v.link.expressions()
        _, dim, _ = mat.shape
        if dim > self.kMaxExternalMvn:
            raise LoweringError("%r: a %dx%d covariance (the batched kernel takes up to %dx%d)"
                                % (v.name, dim, dim, self.kMaxExternalMvn, self.kMaxExternalMvn))
        host_g = dict(identity=lambda x: x, softplus=lambda x: np.logaddexp(0.0, x), sigmoid=lambda x: 1.0 / (1.0 + np.exp(-x)),
                      exp=np.exp, log=np.log, tanh=np.tanh, sqrt=np.sqrt, square=np.square)
        code, mats, memo = [], [], {}
        slot_inputs, uniform_inputs = [], []          # [(IR node, operand row)], [(IR node, uniform entry tuple)]

        def push(ins):
            code.append(ins)
            return len(code) - 1

        def constant(node):
            """numpy value [dim, dim] of a sample-free, parameter-free node (else None)"""
            if node.op == "imm":
                return np.full((dim, dim), float(node.attr))
            m = self.match_uniform(node)
            if m is None:
                return None
            leaf, g, a, b = m
            if leaf.op == "root" and leaf.attr.learnable:
                return None
            raw = np.asarray(leaf.attr.value if leaf.op == "root" else leaf.attr, dtype=np.float64).reshape(leaf.shape[1:])
            return np.broadcast_to(a + b * host_g[g](raw), (dim, dim))

        def emit(node):
            hit = memo.get(node.key)
            if hit is not None:
                return hit
            if any(n not in (1, dim) for n in node.shape[1:]) or node.shape[0] != 1:
                raise LoweringError("%r: a covariance expression mixes shapes (%r in a %dx%d matrix)" % (v.name, node.shape, dim, dim))
            arr = constant(node)
            if arr is not None:
                if np.all(arr == arr.flat[0]):
                    t = push(("IMM", 0, 0, 0, float(arr.flat[0])))
                else:
                    mats.append(np.ascontiguousarray(arr, dtype=np.float32))
                    t = push(("MAT", 0, len(mats) - 1, 0, 0.0))
            elif node.op == "z" or (node.op == "elem" and node.args[0].op == "z"):
                if node.shape != (1, 1, 1):
                    raise LoweringError("%r: a sampled VECTOR inside a covariance expression (only scalars per sample are "
                                        "inputs of the batched kernel)" % v.name)
                child, j = (node, 0) if node.op == "z" else (node.args[0], int(node.attr))
                slot_inputs.append((node, self.slots[child.attr].base + j))
                t = push(("INPUT", 0, ("s", len(slot_inputs) - 1), 0, 0.0))
            elif not node.has_z and self.match_uniform(node) is not None:
                leaf, g, a, b = self.match_uniform(node)                     # a learnable scalar behind its range transform
                if _nelem(leaf.shape) != 1:
                    raise LoweringError("%r: a learnable ARRAY inside a covariance expression (only scalars are inputs of the "
                                        "batched kernel)" % v.name)
                is_param, k0 = self.uniform_entries(leaf, g, a, b)
                uniform_inputs.append((node, self.uni_param[k0]))
                t = push(("INPUT", 0, ("u", len(uniform_inputs) - 1), 0, 0.0))
            elif node.op in BINOP and node.op != "delta":
                x, y = emit(node.args[0]), emit(node.args[1])
                t = push(("BIN", BINOP[node.op], x, y, 0.0))
            elif node.op.startswith("call:") and node.op[5:] in UNOP and node.op[5:] not in ("p2l", "relu", "log1p", "expm1"):
                t = push(("UN", UNOP[node.op[5:]], emit(node.args[0]), 0, 0.0))
            else:
                raise LoweringError("%r: %s inside a covariance expression is not served by the batched kernel" % (v.name, node.op))
            memo[node.key] = t
            return t

        emit(mat)
        if len(slot_inputs) + len(uniform_inputs) > 8:
            raise LoweringError("%r: more than 8 sampled / learnable scalars in a covariance expression" % v.name)
        n_s = len(slot_inputs)
        code = [(k, f, (a[1] if a[0] == "s" else n_s + a[1]) if isinstance(a, tuple) else a, b, imm) for k, f, a, b, imm in code]
        value = self.p_value(v)
        loc = self.from_expr(links["loc"].expr, self.p_value)
        loc_entries, loc_nodes = None, []
        if loc.op == "imm":
            loc_vec = np.full(dim, float(loc.attr))
        else:
            m = self.match_uniform(loc)
            if m is None:
                raise LoweringError("%r: the batched kernel takes a constant or a learnable loc (not one computed from samples)" % v.name)
            leaf, g, a, b = m
            size = _nelem(leaf.shape)
            if size not in (1, dim):
                raise LoweringError("the loc of %r has %d elements, its covariance is %dx%d" % (v.name, size, dim, dim))
            if leaf.op == "root" and leaf.attr.learnable:
                # (by the reference's name-collision rule the prior's loc root is often the posterior's learnable mean)
                is_param, k0 = self.uniform_entries(leaf, g, a, b)
                loc_entries = np.zeros(dim, dtype=UNIFORM_DTYPE)
                for i in range(dim):
                    src, tr, isp, aa, bb = self.uni_param[k0 + (i if size == dim else 0)]
                    loc_entries[i] = (src, tr, isp, 0, aa, bb)
                loc_nodes = [self.mk("elem", (loc,), i if size == dim else 0, (1, 1, 1)) for i in range(dim)] if size == dim else [loc] * dim
                loc_vec = np.zeros(dim)
            else:
                raw = np.asarray(leaf.attr.value if leaf.op == "root" else leaf.attr, dtype=np.float64).reshape(-1)
                loc_vec = np.broadcast_to(a + b * host_g[g](raw), (dim,))
        node = ExternalMvn()
        node.name, node.dim, node.code, node.weight = v.name, dim, code, 1.0
        node.form = given                      # covariance_matrix | scale_tril | precision_matrix (bsvi_mvn_form)
        node.mats = np.stack(mats) if mats else np.zeros((0, dim, dim), np.float32)
        node.loc = np.ascontiguousarray(loc_vec, dtype=np.float32)
        node.slot_inputs = [row for _, row in slot_inputs]
        node.uniform_inputs = np.zeros(len(uniform_inputs), dtype=UNIFORM_DTYPE)
        for k, (_, (src, tr, is_param, a, b)) in enumerate(uniform_inputs):
            node.uniform_inputs[k] = (src, tr, is_param, 0, a, b)
        partners = [n for n, _ in slot_inputs]
        if value.op == "obs":
            data = np.asarray(value.attr._observed_value, dtype=np.float32).reshape(-1)
            if data.size != dim:
                raise LoweringError("the value of %r has %d elements, its covariance is %dx%d" % (v.name, data.size, dim, dim))
            node.value, node.value_row0 = data, 0
        elif value.op == "z" and _nelem(value.shape) == dim:
            node.value, node.value_row0 = None, self.slots[value.attr].base
            partners += [self.mk("elem", (value,), j, (1, 1, 1)) for j in range(dim)]
        elif self.match_uniform(value) is not None:
            # the value is itself [dim] transformed PARAMETERS — the taylor1 program reads the model at the posterior's mean
            # (gradient_estimators.py:47-56), and the mean of Normal(loc, scale) is its learnable loc: uniform entries like a
            # learnable loc's, with the coefficient rows -alpha where a latent value's stand
            leaf, g, a, b = self.match_uniform(value)
            size = _nelem(leaf.shape)
            if not (leaf.op == "root" and leaf.attr.learnable) or size != dim:
                raise LoweringError("%r: the batched kernel takes an observed value, the draw of ONE posterior variable of %d elements "
                                    "or %d learnable values" % (v.name, dim, dim))
            is_param, k0 = self.uniform_entries(leaf, g, a, b)
            node.value_entries = np.zeros(dim, dtype=UNIFORM_DTYPE)
            for i in range(dim):
                src, tr, isp, aa, bb = self.uni_param[k0 + i]
                node.value_entries[i] = (src, tr, isp, 0, aa, bb)
            node.value, node.value_row0 = np.zeros(dim, dtype=np.float32), 0
            partners += [self.mk("elem", (value,), j, (1, 1, 1)) for j in range(dim)]
        elif self.estimator == "taylor1" and value.has_z and _nelem(value.shape) == dim:
            # Taylor1 (`gradient_estimators.py:47-56`) with a posterior whose mean depends on SAMPLED parents (q(f) = Normal(g(z), s),
            # z drawn): the value of the term is g(z) — an expression that differs per sample.  It reaches the batched kernel the
            # way a latent value does, as rows of the draw: a pseudo posterior variable Normal(g(z), 0) behind the posterior's own
            # variables (its draw IS its mean; no entropy, no log q), whose adjoint — the coefficient rows -alpha — runs back
            # through g into z's reparameterised draw and the parameters of g in the program's ordinary reverse sweep.
            if self.pseudo_q:
                raise LoweringError("%r: a second batched multivariate-normal term whose value is a per-sample mean (its rows "
                                    "would lie behind the first term's coefficient rows)" % v.name)

            class _MeanValue:                              # a pseudo posterior variable: Normal(mean expression, 0)
                _pseudo, _mean_value, is_observed = False, True, False

                def __init__(self, name):
                    self.name, self.distribution = name, D.NormalDistribution()

            mean = _MeanValue("%s/mean" % v.name)
            self.slots[mean] = SlotInfo(mean, self.n_slots, value.shape, D.DIST_NORMAL)
            self.n_slots += dim
            self.mean_q.append((mean, [value, self.mk("imm", (), 0.0)], value.shape))
            zval = self.z_node(mean)
            node.value, node.value_row0 = None, self.slots[mean].base
            partners += [self.mk("elem", (zval,), j, (1, 1, 1)) for j in range(dim)]
        else:
            raise LoweringError("%r: the batched kernel takes an observed value, the draw of ONE posterior variable of %d elements or "
                                "(Taylor1) its mean as an expression of sampled parents" % (v.name, dim))
        partners += [n for n, _ in uniform_inputs]
        partners += loc_nodes
        node.loc_entries = loc_entries
        node.n_rows_out = len(partners) + 1
        self.externals.append(node)
        if self.external_mode == "omit":                   # the base program: everything but this term (engine: first launch)
            return []
        # the surrogate: one GIVEN row per coefficient (behind the posterior's real rows), one LINEAR record per row
        node.row0 = self.n_slots

        class _Coefficient:                                # a pseudo posterior variable: its "noise" row holds the coefficient
            _pseudo, is_observed = True, False

            def __init__(self, name):
                self.name, self.distribution = name, D.NormalDistribution()

        class _Surrogate:                                  # what the emission reads of a model variable
            is_observed, b_axis = True, 1

            def __init__(self, name):
                self.name, self.distribution = name, D.LinearSurrogate()

        terms = []
        one, zero = self.mk("imm", (), 1.0), self.mk("imm", (), 0.0)
        for k in range(node.n_rows_out):
            coeff = _Coefficient("%s/coefficient[%d]" % (v.name, k))
            self.slots[coeff] = SlotInfo(coeff, self.n_slots, (1, 1, 1), D.DIST_NORMAL)
            self.n_slots += 1
            self.pseudo_q.append(coeff)
            g = self.mk("z", (), coeff, (1, 1, 1))
            partner = partners[k] if k < len(partners) else one
            terms.append((_Surrogate("%s/surrogate[%d]" % (v.name, k)), partner, [g, zero], (1, 1, 1)))
        return terms

    # ---------------------------------------------------------------- categorical likelihood terms
    def categorical_terms(self, v):
        """An OBSERVED CategoricalVariable of the joint model whose logits are an ordinary (elementwise) link
        (`standard_variables.py:280-299`, `distributions.py:275-311`; the dense path serves `logits = matmul(W, x)`).
        For the label k of a datapoint,  log softmax(l)_k = -log(1 + sum_{c != k} exp(l_c - l_k)) = log sigmoid(-m)  with
        m = log sum_{c != k} exp(l_c - l_k): the log-probability of the outcome 1 under Bernoulli(logits = -m).  One
        Bernoulli record per datapoint, its logits a scalar expression over single elements of the class axis
        (`element_of`); two classes reduce to the Bernoulli likelihood itself.  Labels are data: fixed at lowering time.
        (A LATENT Categorical cannot be pinned: the reference's own `calculate_log_probability` fails on its one-hot
        samples under the installed torch.)"""
        if not (v.is_observed and v.has_observed_value):
            raise LoweringError("Categorical variable %r: only observed labels are lowered on the scalar path" % v.name)
        links = v.link.expressions()
        if "logits" in links:
            logits = self.from_expr(links["logits"].expr, self.p_value)
        else:
            logits = self.mk("call:log", (self.from_expr(links["probs"].expr, self.p_value),))
        B, C, inner = logits.shape
        if inner != 1:
            raise LoweringError("the class axis of %r must be the first element axis ([classes, 1])" % v.name)
        labels = np.asarray(v._observed_value, dtype=np.float64)
        labels = labels.reshape(labels.shape[1], -1)[:, 0]                  # [1, datapoints, ...] -> one label per datapoint
        if np.any(labels != np.round(labels)) or labels.min() < 0 or labels.max() >= C:
            raise LoweringError("labels of %r must be class indices in [0, %d)" % (v.name, C))
        if B not in (1, len(labels)):
            raise LoweringError("%r: %d datapoints of logits against %d labels" % (v.name, B, len(labels)))
        one = self.mk("imm", (), 1.0)

        class _Label:
            def __init__(self, i):
                self.is_observed, self.name, self.distribution = True, "%s[%d]" % (v.name, i), D.BernulliDistribution()

        terms = []
        for b, k in enumerate(labels.astype(int)):
            own = self.element_of(logits, (b, int(k), 0))
            total = None
            for c in range(C):
                if c == k:
                    continue
                e = self.mk("call:exp", (self.mk("sub", (self.element_of(logits, (b, c, 0)), own)),))
                total = e if total is None else self.mk("add", (total, e))
            if total is None:
                continue                                                     # one class: log-probability 0
            terms.append((_Label(b), one, [self.mk("call:neg", (self.mk("call:log", (total,)),))], (1, 1, 1)))
        return terms

    # ---------------------------------------------------------------- driver
    def run(self):
        joint, posterior = self.joint, self.posterior
        q_flat = posterior._flatten()
        names = [v.name for v in q_flat]
        if len(set(names)) != len(names):
            warnings.warn("duplicate variable names in the posterior model; the last one in name order "
                          "wins (reference behaviour, variables.py:79-81)")
        self.q_by_name = {v.name: v for v in q_flat}
        self.q_roots = {v for v in posterior.variables if isinstance(v, RootVariable)}
        p_flat = joint._flatten()
        pnames = [v.name for v in p_flat]
        if len(set(pnames)) != len(pnames):
            warnings.warn("duplicate variable names in the joint model (e.g. the README names y0 'x0'); "
                          "name-based posterior mapping becomes order dependent (reference behaviour)")

        supported = (D.DIST_NORMAL, D.DIST_LOGNORMAL, D.DIST_CAUCHY, D.DIST_LAPLACE, D.DIST_BETA,
                     D.DIST_BINOMIAL, D.DIST_BERNOULLI)

        # -- register learnable parameters in a stable order: posterior group first
        for v in sorted(self.q_roots, key=lambda v: v.name):
            if v.learnable:
                self.param_offset(v.parameter, 0)
        for v in sorted([v for v in joint.flatten() if isinstance(v, RootVariable)], key=lambda v: v.name):
            if v.learnable:
                self.param_offset(v.parameter, 1)

        # -- q: topological order of the random (non-deterministic) variables
        q_random = [v for v in q_flat if isinstance(v, RandomVariable)
                    and getattr(v, "_type", None) != "Deterministic node"]
        order, seen = [], set()

        def visit(v):
            if v in seen:
                return
            seen.add(v)
            for parent in sorted(v.parents, key=lambda x: x.name):
                if isinstance(parent, RandomVariable):
                    visit(parent)
            if isinstance(v, RandomVariable) and getattr(v, "_type", None) != "Deterministic node":
                order.append(v)

        for v in q_random:
            visit(v)

        # -- first pass over q: parameter IR and shapes (slots must exist before children)
        q_nodes = []
        for v in order:
            if v.is_observed:
                raise LoweringError("observed variables inside the posterior are not supported")
            if v.distribution.kind not in supported:
                raise LoweringError("distribution of %r is not supported by the fused kernel yet" % v.name)
            params = self.node_params(v, self.q_value)
            if self.estimator == "taylor1":
                # the entropy is evaluated on the means too (`variables.py:851-855` with samples := means): for a
                # Normal it depends on the scale only, which must therefore not depend on sampled parents
                # (other distributions: every parameter; discrete ones have no analytic entropy in the reference and
                #  fall back to -log q at the sampled value, which the means would change)
                kind = v.distribution.kind
                if kind == D.DIST_NORMAL:
                    ok = not params[1].has_z
                elif kind in (D.DIST_LOGNORMAL, D.DIST_LAPLACE, D.DIST_BETA):
                    ok = not any(p.has_z for p in params)
                else:
                    raise LoweringError("Taylor1 estimator: %r has no analytic entropy (the reference falls back to -log q "
                                        "at the mean, which torch rejects for discrete values)" % v.name)
                if not ok:
                    # the entropy of this node depends on other latents: its term is evaluated on THEIR means
                    # (`variables.py:851-855` with samples := means, `gradient_estimators.py:47-56`) — an entropy-only
                    # record behind the sampling record, with the node's parameters rebuilt on mean_value
                    self.mean_entropy[v] = self.node_params(v, self.mean_value)
            shape = broadcast_shapes3(*[p.shape for p in params])
            if any(self.has_view(p) for p in params):
                if shape != (1, 1, 1):
                    raise LoweringError("the parameters of the posterior variable %r use sum / transpose / [...] and are not "
                                        "one value per sample: only scalar nodes of q take axis views" % v.name)
                params = [self.element_of(p, (0, 0, 0)) for p in params]
            self.slots[v] = SlotInfo(v, self.n_slots, shape, v.distribution.kind)
            self.slot_rank[v] = self.sample_rank(params)
            self.n_slots += self.slots[v].size
            q_nodes.append((v, params, shape))

        # -- p: every random variable of the joint model once (visit-once recursion)
        p_nodes = []
        for v in p_flat:
            if not isinstance(v, RandomVariable) or getattr(v, "_type", None) == "Deterministic node":
                continue
            if v.distribution.kind == D.DIST_EMPIRICAL and v.is_observed:
                continue        # an observed minibatch source: log-probability 0 (ImplicitDistribution, `distributions.py:226-227`)
            if (v.is_observed and not v.has_observed_value and not getattr(v, "has_random_dataset", False)
                    and getattr(v, "_type", None) not in ("Empirical", "Deterministic node")):
                # observed by flag only: drawn once per evaluation (drawn_handle).  Its log-probability at the draw — one number per
                # evaluation, the same for every sample — comes from the reduce node that reads the draw (its last row), below
                self.drawn_handle(v)
                continue
            if v.distribution.kind == D.DIST_MVNORMAL:
                p_nodes.extend(self.mvn_terms(v))
                continue
            if v.distribution.kind == D.DIST_CATEGORICAL:
                p_nodes.extend(self.categorical_terms(v))
                continue
            if v.distribution.kind not in supported:
                raise LoweringError("distribution of %r is not supported by the fused kernel yet" % v.name)
            value = self.p_value(v)
            params = self.node_params(v, self.p_value)
            shape = broadcast_shapes3(value.shape, *[p.shape for p in params])
            elements = self.split_elements(v, value, params, shape)
            if elements is not None:
                p_nodes.extend(elements)
                continue
            p_nodes.append((v, value, params, shape))

        for handle in self.drawn_obs.values():
            nodes = [n for n in self.externals if getattr(n, "kind", "mvn") == "reduce" and n.drawn_name == handle.name]
            if not nodes:
                raise LoweringError("variable %r is observed without a value (drawn per evaluation) and read outside a large reduction: "
                                    "not lowered" % handle.name)
            if nodes[0].logp_node is not None:          # (None: the base program, which makes the draw only)
                p_nodes.append((_SurrogateTerm("%s/log-probability of the draw" % handle.name), self.mk("imm", (), 1.0),
                                [nodes[0].logp_node, self.mk("imm", (), 0.0)], (1, 1, 1)))
        # -- the coefficient rows of batched multivariate-normal terms: GIVEN pseudo-variables behind the posterior's own rows
        q_nodes.extend(self.mean_q)                     # (Taylor1: per-sample means of vector values, rows of the draw like the q's own)
        n_real_q = len(q_nodes)
        for coeff in self.pseudo_q:
            q_nodes.append((coeff, [self.mk("imm", (), 0.0), self.mk("imm", (), 1.0)], self.slots[coeff].shape))

        # -- weights from the [N, B] mean rule
        term_b = []
        for v, params, shape in q_nodes[:n_real_q]:      # (GIVEN rows have no term of their own: a vector of them — the reduce node's — does not count)
            term_b.append(shape[0])
        for v, value, params, shape in p_nodes:
            term_b.append(1 if v.is_observed else getattr(v, "b_axis", shape[0]))
        bmax = max(term_b) if term_b else 1
        for b in term_b:
            if b not in (1, bmax):
                raise LoweringError("datapoint axes %r of the ELBO terms cannot be broadcast" % (sorted(set(term_b)),))
        if self.estimator in ("blackbox", "importance") and bmax != 1:
            raise LoweringError("BlackBox estimator / importance weights with a datapoint axis on latent terms are "
                                "not lowered yet")

        def weight(b_term):
            return 1.0 if b_term == 1 else 1.0 / bmax

        # -- shared sample-dependent sub-expressions become derived slots (computed once per sample)
        for v, params, shape in q_nodes:
            self.count_uses(params)
            if v in self.mean_entropy:
                self.count_uses(self.mean_entropy[v])
        for v, value, params, shape in p_nodes:
            self.count_uses([value] + params)
        self.derived_nodes = {k for k, c in self.use_count.items() if c >= 2}
        self.n_latent = self.n_slots
        n_derived_upper = 0
        for k in self.derived_nodes:
            n_derived_upper += _nelem(self.ir_cache[k].shape)
        self.temp_base = self.n_latent + n_derived_upper

        # -- emit q records
        for v, params, shape in q_nodes:
            dist = v.distribution
            slot = self.slots[v]
            self.ensure_derived(params)
            w = weight(shape[0])
            flags = F_SAMPLE
            w_lp = 0.0
            w_ent = w
            if getattr(v, "_pseudo", False):
                # its row of the noise tensor IS its value (a coefficient the batched kernel wrote); no term of its own
                self.for_each_record(shape, params, False,
                                     lambda: self.emit_node(dist.kind, F_SAMPLE | F_GIVEN, params, slot=slot, w_lp=0.0, w_ent=0.0))
                continue
            if getattr(v, "_mean_value", False):
                # Normal(mean expression, 0): the row of the draw carries the mean; no term of its own
                self.for_each_record(shape, params, False,
                                     lambda: self.emit_node(dist.kind, F_SAMPLE, params, slot=slot, w_lp=0.0, w_ent=0.0))
                continue
            if self.estimator == "importance":
                flags |= F_WF | F_GIVEN        # value supplied, log q accumulated, no entropy term
                w_ent = 0.0
            elif v in self.mean_entropy:
                # Taylor1, entropy on the parents' means: the sampling record carries no term, the record behind it
                # reads the same distribution's parameters evaluated on the means and adds w * H
                mean_params = self.mean_entropy[v]
                self.ensure_derived(mean_params)
                self.for_each_record(shape, params, False,
                                     lambda: self.emit_node(dist.kind, F_SAMPLE, params, slot=slot, w_lp=0.0, w_ent=0.0))
                self.for_each_record(shape, mean_params, False,
                                     lambda: self.emit_node(dist.kind, F_ENT, mean_params, value=self.z_node(v), w_lp=0.0, w_ent=w))
                continue
            elif dist.has_analytic_entropy:
                flags |= F_ENT
            else:
                flags |= F_LOGP
                w_lp = -w                      # entropy fallback -log q (variables.py:161-162)
            if self.estimator == "blackbox":
                flags |= F_WF
            self.for_each_record(shape, params, False,
                                 lambda: self.emit_node(dist.kind, flags, params, slot=slot, w_lp=w_lp, w_ent=w_ent))

        # -- emit p records
        for v, value, params, shape in p_nodes:
            self.ensure_derived([value] + params)
            # a model log-prob term has a constant weight and nothing depends on its value: the
            # kernel finishes it (value and adjoints) in the forward sweep
            w = 1.0 if v.is_observed else weight(getattr(v, "b_axis", shape[0]))
            self.for_each_record(shape, [value] + params, True,
                                 lambda: self.emit_node(v.distribution.kind, F_LOGP, params, value=value, w_lp=w))

        return self.finish(bmax)

    # ---------------------------------------------------------------- sampling programs
    def run_sampler(self, input_values):
        """Ancestral sampling program: the posterior's variables first (if a posterior
        is attached: `ProbabilisticModel._get_posterior_sample`, variables.py:796-805, 903-907), then
        every variable of the model that did not receive a value from the posterior by name —
        `ProbabilisticModel._get_sample(observed=False)`, variables.py:732-742, 527-570.
        Deterministic nodes get a slot too so that their values are part of the result."""
        joint, posterior = self.joint, self.posterior
        self.q_by_name, self.q_roots = {}, set()
        if posterior is not None:
            self.q_by_name = {v.name: v for v in posterior._flatten()}
            self.q_roots = {v for v in posterior.variables if isinstance(v, RootVariable)}
        given = {}
        for var, value in (input_values or {}).items():
            arr = np.asarray(value, dtype=np.float32) if not hasattr(value, "detach") else value.detach().cpu().numpy()
            if arr.ndim >= 1 and arr.shape[0] > 1:
                raise LoweringError("per-sample input_values are not supported by sampling programs yet")
            given[var] = np.ascontiguousarray(arr.reshape(arr.shape[1:]) if arr.ndim > 1 else arr.reshape(1))
        self.outputs = []          # (variable, slot) in sampling order

        def topo(variables):
            order, seen = [], set()

            def visit(v):
                if v in seen or not isinstance(v, RandomVariable):
                    return
                seen.add(v)
                for parent in sorted(v.parents, key=lambda x: x.name):
                    visit(parent)
                order.append(v)

            for v in sorted(variables, key=lambda x: x.name):
                visit(v)
            return order

        def const_leaf(var):
            arr = given[var]
            if arr.size == 1:
                return self.mk("imm", (), float(arr.reshape(-1)[0]))
            return self.mk("carr", (), arr, canonical_elem_shape((1,) + arr.shape) if arr.ndim < 3 else canonical_elem_shape(arr.shape))

        def q_ctx(var):
            if var in given:
                return const_leaf(var)
            if var in self.slots:
                return self.z_node(var)
            return self.q_value(var)

        def p_ctx(var):
            if var in given:
                return const_leaf(var)
            if var in self.slots:
                return self.z_node(var)
            if var.name in self.q_by_name:
                return q_ctx(self.q_by_name[var.name])
            if isinstance(var, RootVariable):
                return self.root_node(var)
            raise LoweringError("variable %r is used before it is sampled" % var.name)

        supported = (D.DIST_NORMAL, D.DIST_LOGNORMAL, D.DIST_CAUCHY, D.DIST_LAPLACE, D.DIST_BETA,
                     D.DIST_BINOMIAL, D.DIST_BERNOULLI, D.DIST_DETERMINISTIC)
        plan = []
        if posterior is not None:
            plan += [(v, q_ctx) for v in topo(posterior._flatten()) if v not in given]
        mapped = set(self.q_by_name)
        plan += [(v, p_ctx) for v in topo(joint._flatten()) if v not in given and v.name not in mapped]
        for v, ctx in plan:
            if v.distribution.kind not in supported:
                raise LoweringError("distribution of %r is not supported by the fused kernel yet" % v.name)
            params = self.node_params(v, ctx)
            shape = broadcast_shapes3(*[p.shape for p in params])
            slot = SlotInfo(v, self.n_slots, shape, v.distribution.kind)
            self.n_slots += slot.size
            self.n_latent = self.n_slots
            plan_item = (v, params, shape, slot)
            self.slots[v] = slot
            self.slot_rank[v] = self.sample_rank(params)
            self.outputs.append(plan_item)
        self.temp_base = self.n_slots          # no derived slots in sampling programs
        self.derived_nodes = set()
        for v, params, shape, slot in self.outputs:
            try:
                self.for_each_record(shape, params, False,
                                     lambda: self.emit_node(v.distribution.kind, F_SAMPLE, params, slot=slot))
            except LoweringError as err:
                if any(self.has_view(p) for p in params) and "vsum" in str(err):
                    raise LoweringError("sampling %r: its parameters hold a reduction that only the ELBO programs evaluate (the reduce node, "
                                        "DESIGN 4.1) — sample the posterior's own variables with model.posterior_model.get_sample(...)" % v.name) from err
                raise
        prog = self.finish(1)
        prog.outputs = [(v, slot) for v, _, _, slot in self.outputs]
        return prog

    def fill_parameter_tables(self, prog, uni, n_up):
        """CSR map parameter -> uniform entries, activity mask and optimizer groups"""
        prog.n_params = self.n_params
        prog.n_uniform_grad = n_up
        ptr = np.zeros(self.n_params + 1, dtype=np.uint32)
        src = uni["src"][:n_up].astype(np.int64)
        np.add.at(ptr, src + 1, 1)
        ptr = np.cumsum(ptr).astype(np.uint32)
        prog.param_uniform_ptr = ptr
        prog.param_uniform_idx = np.argsort(src, kind="stable").astype(np.uint32)
        prog.parameters = list(self.parameters)
        active = np.zeros(self.n_params, dtype=np.uint8)
        active[src] = 1
        prog.param_active = active
        group = np.zeros(self.n_params, dtype=np.uint8)
        for p, off, size, g in self.parameters:
            group[off:off + size] = g
        prog.param_group = group

    def uniform_table(self):
        n_up = len(self.uni_param)
        uni = np.zeros(n_up + len(self.uni_const), dtype=UNIFORM_DTYPE)
        for k, (src, tr, is_param, a, b) in enumerate(self.uni_param + self.uni_const):
            uni[k] = (src, tr, is_param, 0, a, b)
        return uni, n_up

    def finish(self, bmax):
        prog = Program()
        prog.estimator = self.estimator
        n_up = len(self.uni_param)
        uni = np.zeros(n_up + len(self.uni_const), dtype=UNIFORM_DTYPE)
        for k, (src, tr, is_param, a, b) in enumerate(self.uni_param + self.uni_const):
            uni[k] = (src, tr, is_param, 0, a, b)
        n_uni = len(uni)
        # flatten the records into ONE instruction stream.  A record that is a single instruction
        # over a single element (the overwhelmingly common case) is that instruction; anything
        # else is bracketed by REC_BEGIN / REC_END pseudo-instructions that carry the loop
        # extent and the temp range, so that both sweeps find record boundaries in the stream.
        def no_alias(ins, n_elems):
            """True when the adjoint cells of the instruction's operands are pairwise distinct for
            every element of the record's loop (constants / observed data have no adjoint)."""
            spans = []
            for kind, index, stride in ins[1:6]:
                if kind == K_Z or kind == K_U:
                    spans.append((kind, index, index + stride * (n_elems - 1)))
            for i in range(len(spans)):
                for j in range(i + 1, len(spans)):
                    (k0, a0, b0), (k1, a1, b1) = spans[i], spans[j]
                    if k0 == k1 and a0 <= b1 and a1 <= b0:
                        return False
            return True

        def assemble(records, own_terms=True):
            """records -> (code words, record table).  own_terms=False: the posterior's sampling nodes keep sampling
            but contribute no entropy / log q term of their own (a share other than the first of a split program)."""
            words, recs_out = [], []
            for rec in records:
                (b, e, n_elems, n_temps, sink), lo = rec[:5], (rec[5] if len(rec) > 5 else 0)
                body = []
                for ins in self.code[b:e]:
                    imm0, imm1 = ins[6], ins[7]
                    if not own_terms and (ins[0] & 0xFF) in (OP["NAFF"], OP["NODE"]) and ((ins[0] >> 8) & F_SAMPLE):
                        imm0 = imm1 = _fbits(0.0)
                    if lo:      # an element sub-range of the record: every operand starts `lo` strides further
                        ins = [ins[0]] + [(k, i + s * lo, s) for (k, i, s) in ins[1:6]] + [imm0, imm1]
                    body.append([ins[0] | ((R_NOALIAS << 24) if no_alias(ins, n_elems) else 0)]
                                + [encode_operand(o, n_up, n_uni) for o in ins[1:6]] + [imm0, imm1])
                rflag = (R_SINK if sink else 0) << 24
                if len(body) == 1 and n_elems == 1:
                    body[0][0] |= rflag
                    recs_out.append((len(words), len(words) + 1, n_elems, self.temp_base, n_temps, int(sink)))
                    words.extend(body)
                else:
                    bracket = [len(body), n_elems, self.temp_base, n_temps, 0, 0, 0]
                    words.append([OP["REC_BEGIN"] | rflag] + bracket)
                    recs_out.append((len(words), len(words) + len(body), n_elems, self.temp_base, n_temps, int(sink)))
                    words.extend(body)
                    words.append([OP["REC_END"] | rflag] + bracket)
            code = np.array(words, dtype=np.uint32).reshape(-1, 8) if words else np.zeros((0, 8), np.uint32)
            recs = np.zeros(len(recs_out), dtype=RECORD_DTYPE)
            for i, r in enumerate(recs_out):
                recs[i] = r
            return code, recs

        code, recs = assemble(self.records)
        # Shares of the program for the multi-workgroup persistent trainer (engine.train): every share
        # samples the posterior (all non-sink records) but evaluates only every V-th model log-prob record; the
        # estimator value and all adjoints are linear in those records, so the shares' partial sums add up to the
        # full program's.  Pathwise only: BlackBox multiplies per-sample totals.
        sinks = [i for i, r in enumerate(self.records) if r[4]]
        # units of work: the elements of the sink records (a record over a datapoint / vector axis is split by elements:
        # share v takes a contiguous sub-range, operands shifted by that many strides)
        work = sum(self.records[i][2] for i in sinks)
        all_records = list(self.records)

        def parts_of(V):
            parts = []
            for v in range(V):
                records, single = [], 0
                for i, r in enumerate(all_records):
                    if not r[4]:
                        records.append(r)
                    elif r[2] == 1:
                        if single % V == v:
                            records.append(r)
                        single += 1
                    else:
                        lo, hi = (v * r[2]) // V, ((v + 1) * r[2]) // V
                        if hi > lo:
                            records.append((r[0], r[1], hi - lo, r[3], r[4], lo))
                parts.append(assemble(records, own_terms=(v == 0)))
            return parts

        # (assembled when first asked for: the 23 streams of V = 2, 3, 4, 6, 8 were 40 % of the lowering of the README model, and a
        #  single-workgroup launch asks for none of them)
        splits = [V for V in (2, 3, 4, 6, 8) if work >= 2 * V] if (self.estimator == "pathwise" and work >= 6) else []
        prog.shares = _LazyShares(parts_of, splits)
        prog.uniform, prog.records, prog.code = uni, recs, code
        prog.consts = np.concatenate(self.consts) if self.consts else np.zeros(0, np.float32)
        prog.obs = np.concatenate(self.obs) if self.obs else np.zeros(0, np.float32)
        # minibatch observations (scalar-path f-1): which stretches of the observation buffer are rows of which dataset.  Sources that share
        # a RandomIndices variable share a draw (`group`); an EmpiricalVariable with its own batch_size draws for itself.
        prog.minibatches = []
        groups = {}
        for handle in self.minibatch_obs.values():
            if id(handle) not in self.obs_index:
                continue                           # (lowered but never read: no stretch of the buffer)
            key = id(handle.indices) if handle.indices is not None else id(handle)
            group = groups.setdefault(key, len(groups))
            prog.minibatches.append(dict(name=handle.name, offset=int(self.obs_index[id(handle)]), batch=handle.batch,
                                         row=int(math.prod(int(d) for d in handle.dataset.shape[1:])), dataset=handle.dataset,
                                         dataset_size=int(handle.dataset.shape[0]), group=group,
                                         indices_name=handle.indices.name if handle.indices is not None else handle.name))
        prog.n_noise = self.n_latent
        prog.n_derived = self.temp_base - self.n_latent
        prog.n_temps = self.max_temps
        prog.n_slots = self.temp_base + self.max_temps
        self.fill_parameter_tables(prog, uni, n_up)
        prog.slots = dict(self.slots)
        prog.slot_by_name = {s.name: s for s in self.slots.values()
                             if not getattr(s.var, "_pseudo", False) and not getattr(s.var, "_mean_value", False)}
        # batched multivariate-normal terms (mvn_external): their descriptions, and how many noise rows are the posterior's own
        prog.externals = list(self.externals)
        prog.n_real_noise = self.n_latent - sum(self.slots[c].size for c in self.pseudo_q)      # (a reduce node's GIVEN rows are vectors)
        prog.bmax = bmax
        prog.op_count = len(code)
        return prog


def _nelem(shape):
    """number of elements of a shape (np.prod of a small tuple costs 5 us a call; the lowering makes hundreds)"""
    return int(math.prod(int(d) for d in shape))


class _LazyShares(dict):
    """Program.shares: V -> [(code, records)] * V, every split assembled at its first use."""

    def __init__(self, build, keys):
        super().__init__((k, None) for k in keys)
        self._build = build

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if v is None:
            v = self._build(k)
            dict.__setitem__(self, k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class ExternalMvn:
    """A multivariate-normal term evaluated by the batched kernel of the library (lowering.mvn_external): the description
    `native.mvn_desc` turns into a bsvi_mvn_desc, and where its rows live in the program's noise tensor."""
    name = dim = code = mats = loc = value = uniform_inputs = slot_inputs = weight = loc_entries = value_entries = None
    value_row0 = row0 = n_rows_out = 0


def lower_sampler(model, posterior_model=None, input_values=None):
    """Compile an ancestral-sampling program for `model` (optionally conditioned on its posterior)."""
    return _Lowering(model, posterior_model, "pathwise").run_sampler(input_values)


def lower(joint_model, posterior_model=None, estimator="pathwise", external="inject"):
    """Compile a (joint, posterior) pair for the fused ELBO kernel.  external="omit": the BASE program of a model with
    batched multivariate-normal terms (`Program.externals`) — the same program without those terms and their surrogate
    rows, launched first to make the draw the batched kernel reads (same parameter layout, same noise rows)."""
    if posterior_model is None:
        joint_model.check_posterior_model()
        posterior_model = joint_model.posterior_model
    if not isinstance(joint_model, ProbabilisticModel) or not isinstance(posterior_model, ProbabilisticModel):
        raise ValueError("lower() expects probabilistic models")
    low = _Lowering(joint_model, posterior_model, estimator)
    low.external_mode = external
    return low.run()
