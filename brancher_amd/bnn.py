"""
Bayesian neural networks on the dense-link path: lowering and engine for graphs whose likelihood goes through a CHAIN of
``BF.matmul`` links with latent weight matrices and latent biases,

    hidden = BF.tanh(BF.matmul(weights1, x) + b1)
    logits = BF.matmul(weights2, hidden) + b2
    k      = CategoricalVariable(logits=logits, name="k");  k.observe(labels)
    q:       NormalVariable(loc, scale, <same name>, learnable=True) for weights1, b1, weights2, b2

— the reference's `tests/test_MNIST_bayesian_neural_network.py:20-60` (any depth, any of tanh / relu / sigmoid / softplus
between the layers, biases optional).  `dense.lower_dense` serves ONE latent matrix without a bias; this module serves the
rest of the family through `bsvi_bnn_*` (include/bsvi.h, csrc/bnn_kernel.inc).  The reference's semantics are kept as on the
dense path: priors and posteriors are matched by NAME, auto-created roots `<var>_<arg>` collide by name, no
minibatch rescaling of the likelihood (`variables.py:849` TODO).
"""
import ctypes as C

import numpy as np
import torch

from brancher_amd import distributions as D
from brancher_amd import lowering, native
from brancher_amd.lowering import LoweringError, _Lowering
from brancher_amd.native import OUT_HEADER, BnnArgs, BnnDesc, BnnLayer
from brancher_amd.variables import RandomVariable, RootVariable

LIK_CATEGORICAL, LIK_BERNOULLI = 0, 1
ACTIVATIONS = dict(tanh=1, relu=2, sigmoid=3, softplus=4)
NO_BIAS = 0xFFFFFFFF


class BnnProgram:
    """What `lower_bnn` extracts from the graph."""
    estimator = "pathwise"

    def summary(self):
        return dict(kind="bnn", layers=[(l["rows"], l["cols"], l["activation"]) for l in self.layers], n_rows=self.n_rows,
                    dataset_size=self.dataset_size, batch_size=self.batch_size, n_params=self.n_params,
                    likelihood=("categorical", "bernoulli")[self.likelihood], latents=[t["name"] for t in self.tensors])


def _is_random(v):
    return isinstance(v, RandomVariable) and getattr(v, "_type", None) != "Deterministic node"


def _chain(e):
    """logits expression -> ([(W variable, bias variable | None, activation of the layer's output | None)] bottom up, x variable)"""
    def is_call(x, names):
        return x.op == "call" and isinstance(x.attr[0], str) and x.attr[0] in names

    def affine(x):
        if x.op == "add":
            a, b = x.args
            if is_call(a, ("matmul",)) and b.op == "var":
                return a, b.attr
            if is_call(b, ("matmul",)) and a.op == "var":
                return b, a.attr
            raise LoweringError("bnn path: a layer must be BF.matmul(weights, input) [+ bias]")
        if is_call(x, ("matmul",)):
            return x, None
        raise LoweringError("bnn path: a layer must be BF.matmul(weights, input) [+ bias]")

    mm, bias = affine(e)
    if len(mm.args) != 2 or mm.args[0].op != "var":
        raise LoweringError("bnn path: the first matmul operand must be the weight variable")
    W, inner = mm.args[0].attr, mm.args[1]
    if inner.op == "var":
        return [[W, bias, None]], inner.attr
    if is_call(inner, tuple(ACTIVATIONS)) and len(inner.args) == 1 and not inner.attr[1]:
        below, x = _chain(inner.args[0])
        below[-1][2] = inner.attr[0]
        return below + [[W, bias, None]], x
    raise LoweringError("bnn path: between two layers stands one of BF.%s" % " / BF.".join(ACTIVATIONS))


def lower_bnn(joint, posterior, estimator="pathwise"):
    # "taylor1" (gradient_estimators.py:47-56): f at the posterior's analytic means — every latent of the network is a mean-field Normal,
    # whose mean is its loc (distributions.py:137 through torch), so the program is the Pathwise one evaluated on the draw eps = 0
    # (CompiledBnn supplies it); the entropy of a Normal does not depend on the draw.  The reference tiles the means to number_samples
    # identical rows and averages: kept (N identical samples), so that the value rounds as the reference's does.
    if estimator not in ("pathwise", "blackbox", "taylor1"):
        raise LoweringError("the bnn path implements the Pathwise, BlackBox and Taylor1 estimators")
    L = _Lowering(joint, posterior, estimator)
    q_flat = posterior._flatten()
    L.q_by_name = {v.name: v for v in q_flat}
    L.q_roots = {v for v in posterior.variables if isinstance(v, RootVariable)}
    for v in sorted(L.q_roots, key=lambda v: v.name):
        if v.learnable:
            L.param_offset(v.parameter, 0)
    for v in sorted([v for v in joint.flatten() if isinstance(v, RootVariable)], key=lambda v: v.name):
        if v.learnable:
            L.param_offset(v.parameter, 1)
    q_random = [v for v in q_flat if _is_random(v)]
    if not q_random or any(v.distribution.kind != D.DIST_NORMAL for v in q_random):
        raise LoweringError("bnn path: the posterior must be mean-field Normal variables")
    p_random = [v for v in joint._flatten() if _is_random(v)]
    liks = [v for v in p_random if v.distribution.kind in (D.DIST_CATEGORICAL, D.DIST_BINOMIAL, D.DIST_BERNOULLI)]
    if len(liks) != 1:
        raise LoweringError("bnn path: expected one matmul likelihood")
    k = liks[0]
    if not k.is_observed or not k.has_random_dataset:
        raise LoweringError("bnn path: the likelihood must be observed through an EmpiricalVariable of labels")
    links = k.link.expressions()
    if "logits" not in links:
        raise LoweringError("bnn path: the likelihood must be parameterised by logits")
    chain, x_var = _chain(links["logits"].expr)
    latents = []
    for W, b, _ in chain:
        latents.append(W)
        if b is not None:
            latents.append(b)
    names = [v.name for v in latents]
    if len(set(names)) != len(names):
        raise LoweringError("bnn path: a latent tensor is used by two layers")
    others = [v for v in p_random if v is not k and v not in latents and v.distribution.kind != D.DIST_EMPIRICAL]
    if others or sorted(names) != sorted(v.name for v in q_random):
        raise LoweringError("bnn path: every latent of the network needs a Normal posterior of the same name, and nothing else may be latent")
    for v in latents:
        if not _is_random(v) or v.distribution.kind != D.DIST_NORMAL:
            raise LoweringError("bnn path: weights and biases must be Normal variables")
    labels_var = k.dataset

    def minibatch_source(v, what):
        if getattr(v, "_type", None) != "Empirical" or not v.is_observed:
            raise LoweringError("bnn path: %s must be an observed EmpiricalVariable" % what)
        exprs = v.link.expressions()
        ds = exprs["dataset"].expr
        if ds.op != "var" or not isinstance(ds.attr, RootVariable) or "indices" not in exprs:
            raise LoweringError("bnn path: %s must index an array dataset through a RandomIndices variable" % what)
        ind = exprs["indices"].expr
        from brancher_amd.standard_variables import RandomIndices
        if ind.op != "var" or not isinstance(ind.attr, RandomIndices):
            raise LoweringError("bnn path: %s must be indexed by a RandomIndices variable" % what)
        return np.asarray(ds.attr.value, dtype=np.float32), ind.attr

    X, ind_x = minibatch_source(x_var, "x")
    Y, ind_y = minibatch_source(labels_var, "labels")
    if ind_x is not ind_y:
        raise LoweringError("bnn path: x and labels must share one RandomIndices variable")
    DS = X.shape[1]
    Xm, Ym = X.reshape(DS, -1), Y.reshape(-1)
    if Ym.shape[0] != DS:
        raise LoweringError("bnn path: dataset sizes of x and labels differ")
    P = Xm.shape[1]
    if P % 4:
        raise LoweringError("bnn path: the number of features must be a multiple of 4")

    def row_params(var, ctx):
        out = []
        for node in L.node_params(var, ctx):
            m = L.match_uniform(node)
            if m is None:
                raise LoweringError("bnn path: the parameters of %r must be constants or parameter transforms" % var.name)
            leaf, g, a, b = m
            is_param, k0 = L.uniform_entries(leaf, g, a, b)
            out.append((is_param, k0, int(np.prod(leaf.shape)), leaf.shape))
        return out

    # the latent vector: weights1 first (bsvi_bnn_desc), then the other tensors layer by layer
    order = [chain[0][0]] + ([chain[0][1]] if chain[0][1] is not None else [])
    for W, b, _ in chain[1:]:
        order += [W] + ([b] if b is not None else [])
    tensors, row0, entries = [], 0, []
    for v in order:
        q = L.q_by_name[v.name]
        ql, qs = row_params(q, L.q_value)
        pl, ps = row_params(v, L.p_value)
        shape = tuple(int(s) for s in ql[3])
        if len(shape) != 3 or shape[0] != 1:
            raise LoweringError("bnn path: %r must be a matrix [rows, cols] or a column [rows, 1]" % v.name)
        size = shape[1] * shape[2]
        for ent in (ql, qs, pl, ps):
            if ent[2] not in (1, size):
                raise LoweringError("bnn path: a parameter of %r has an unsupported shape" % v.name)
        tensors.append(dict(name=v.name, row0=row0, rows=shape[1], cols=shape[2], size=size))
        entries.append((ql, qs, pl, ps))
        row0 += size
    R = row0
    by_name = {t["name"]: t for t in tensors}
    layers, width = [], P
    for W, b, act in chain:
        t = by_name[W.name]
        if t["cols"] != width:
            raise LoweringError("bnn path: %r is [%d, %d] but its input has %d rows" % (W.name, t["rows"], t["cols"], width))
        bias0 = NO_BIAS
        if b is not None:
            tb = by_name[b.name]
            if (tb["rows"], tb["cols"]) != (t["rows"], 1):
                raise LoweringError("bnn path: bias %r must be a column [%d, 1]" % (b.name, t["rows"]))
            bias0 = tb["row0"]
        layers.append(dict(rows=t["rows"], cols=t["cols"], weight_row0=t["row0"], bias_row0=bias0,
                           activation=ACTIVATIONS[act] if act else 0, weights=W.name, bias=b.name if b is not None else None))
        width = t["rows"]
    lik = LIK_CATEGORICAL if k.distribution.kind == D.DIST_CATEGORICAL else LIK_BERNOULLI
    if lik == LIK_BERNOULLI:
        if width != 1:
            raise LoweringError("bnn path: a Bernoulli/Binomial likelihood needs a single output")
        if k.distribution.kind == D.DIST_BINOMIAL:
            tc = L.match_uniform(L.from_expr(links["total_count"].expr, L.p_value))
            if tc is None or tc[0].op != "root" or float(np.asarray(tc[0].attr.value).reshape(-1)[0]) != 1.0:
                raise LoweringError("bnn path: Binomial likelihood supports total_count = 1 only")

    uni, n_up = L.uniform_table()
    prog = BnnProgram()
    prog.estimator = estimator
    L.fill_parameter_tables(prog, uni, n_up)
    row_uniform = np.zeros((4, R), dtype=np.uint32)
    for t, ents in zip(tensors, entries):
        for q, (is_param, k0, size, _) in enumerate(ents):
            base = k0 if is_param else n_up + k0
            row_uniform[q, t["row0"]:t["row0"] + t["size"]] = base + (np.arange(t["size"]) if size > 1 else 0)
    prog.uniform = uni
    prog.consts = np.concatenate(L.consts) if L.consts else np.zeros(0, np.float32)
    prog.row_uniform = row_uniform
    prog.layers, prog.tensors, prog.n_rows = layers, tensors, R
    prog.n_features, prog.n_classes, prog.dataset_size, prog.batch_size = P, width, DS, int(ind_x.batch_size)
    prog.likelihood = lik
    prog.dataset = np.ascontiguousarray(Xm, dtype=np.float32)
    prog.labels = np.ascontiguousarray(Ym, dtype=np.float32)
    prog.indices_name = ind_x.name
    prog.lik_weight, prog.prior_weight, prog.entropy_weight = 1.0, 1.0, 1.0
    prog.n_noise = R
    prog.bmax = 1
    return prog


class CompiledBnn:
    """Engine for a Bayesian neural network; same surface as dense.CompiledDense."""

    def data_path(self):
        """which matrix-core path serves the two products with the minibatch: "bf16x3" when every dataset value is exactly a
        bf16 number (three bf16 MFMAs on the exact pieces of the f32 operand), else "f32" (the f32-input MFMA kernels)"""
        return "bf16x3" if self._exact_data else "f32"

    def __init__(self, joint_model, posterior_model, estimator="pathwise", device=None, program=None):
        from brancher_amd import engine
        self.device = device or engine._device()
        self.program = program if program is not None else lower_bnn(joint_model, posterior_model, estimator)
        p = self.program
        lib = native.load()
        if lib.bsvi_device_count() < 1:
            raise native.NativeError("no MI355X / HIP device visible: the engine cannot run (no CPU fallback)")
        self.lib = lib
        layers = (BnnLayer * len(p.layers))(*[BnnLayer(rows=l["rows"], cols=l["cols"], weight_row0=l["weight_row0"],
                                                        bias_row0=l["bias_row0"], activation=l["activation"]) for l in p.layers])
        self._keep = dict(uniform=np.ascontiguousarray(p.uniform), consts=np.ascontiguousarray(p.consts, dtype=np.float32),
                          ptr=np.ascontiguousarray(p.param_uniform_ptr, dtype=np.uint32),
                          idx=np.ascontiguousarray(p.param_uniform_idx, dtype=np.uint32),
                          rows=np.ascontiguousarray(p.row_uniform, dtype=np.uint32), dataset=p.dataset, labels=p.labels, layers=layers)
        k = self._keep
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None
        d = BnnDesc(abi_version=native.ABI_VERSION, n_params=p.n_params, n_consts=k["consts"].size, n_uniform=len(k["uniform"]),
                    n_uniform_grad=p.n_uniform_grad, n_layers=len(p.layers), n_rows=p.n_rows, n_features=p.n_features,
                    dataset_size=p.dataset_size, batch_size=p.batch_size, likelihood=p.likelihood,
                    estimator=lowering.EST[getattr(p, "estimator", "pathwise")], lik_weight=p.lik_weight,
                    prior_weight=p.prior_weight, entropy_weight=p.entropy_weight, layers=C.cast(layers, C.c_void_p),
                    row_uniform=ptr(k["rows"]), uniform=ptr(k["uniform"]), consts=ptr(k["consts"]),
                    param_uniform_ptr=ptr(k["ptr"]), param_uniform_idx=ptr(k["idx"]), dataset=ptr(k["dataset"]), labels=ptr(k["labels"]))
        handle = C.c_void_p()
        native.check(lib.bsvi_bnn_create(C.byref(d), C.byref(handle)))
        self.handle = handle
        self._exact_data = bool(lib.bsvi_bnn_exact_data(handle))
        dev = self.device
        self.n_params = p.n_params
        theta = np.zeros(p.n_params, dtype=np.float32)
        for par, off, size, _ in p.parameters:
            theta[off:off + size] = par.numpy().reshape(-1)
        self.params = _engine.broadcast_from_rank0(torch.from_numpy(theta).to(dev))
        self.out = torch.zeros(OUT_HEADER + max(p.n_params, 1), device=dev)
        active = np.ascontiguousarray(p.param_active, dtype=np.uint8)
        group = p.param_group
        first_group = 0 if np.any(active[group == 0]) else 1
        self.mask_all = torch.from_numpy(active.copy()).to(dev)
        self.mask_first = torch.from_numpy((active * (group == first_group)).astype(np.uint8)).to(dev)
        self._workspaces = {}
        self.iteration = 0
        self.grads_valid = False
        self.last_mode = None
        for par, off, size, _ in p.parameters:
            par.bind(self, off)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.bsvi_bnn_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ParameterStore protocol
    def read_params(self, offset, size):
        return self.params[offset:offset + size].detach().cpu().numpy()

    def write_params(self, offset, values):
        self.params[offset:offset + values.size] = torch.from_numpy(np.ascontiguousarray(values)).to(self.device)

    def read_grads(self, offset, size):
        if not self.grads_valid:
            return None
        o = OUT_HEADER + offset
        return self.out[o:o + size].detach().cpu().numpy()

    def workspace(self, n_local):
        ws = self._workspaces.get(n_local)
        if ws is None:
            # (zeroed once: the pad columns of the product operands are never written)
            ws = torch.zeros(int(self.lib.bsvi_bnn_workspace_bytes(self.handle, n_local)), dtype=torch.uint8, device=self.device)
            self._workspaces[n_local] = ws
        return ws

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _args(self, n_local, n_global, base, noise=None, indices=None, seed=None, offset=0, noise_out=None,
              indices_out=None, fvalue_out=None, logq_out=None, f_weight=None, q_weight=None):
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        if getattr(self.program, "estimator", "pathwise") == "taylor1":
            # f at the posterior's means: the draw eps = 0, whatever was drawn (the reference draws too and reads the means:
            # gradient_estimators.py:50-55) — a caller's noise (parity tests replay the reference's) is not read
            noise = self._zero_noise(n_local)
        if not isinstance(seed, int) or seed is True:      # (train() resolves the call's seed ONCE and hands the integer down)
            seed = _engine.shared_seed(seed, self.device)
        return BnnArgs(params_dev=ptr(self.params), noise_dev=ptr(noise), indices_dev=ptr(indices), seed=seed, offset=int(offset),
                       n_samples_local=n_local, n_samples_global=n_global, sample_base=base, out_dev=ptr(self.out),
                       noise_out_dev=ptr(noise_out), indices_out_dev=ptr(indices_out), fvalue_out_dev=ptr(fvalue_out),
                       logq_out_dev=ptr(logq_out), workspace_dev=ptr(self.workspace(n_local)), stream=self._stream(),
                       f_weight_dev=ptr(f_weight), q_weight_dev=ptr(q_weight))

    def _zero_noise(self, n_local):
        z = self.__dict__.setdefault("_zeros", {}).get(n_local)
        if z is None:
            z = self._zeros[n_local] = torch.zeros((self.program.n_noise, n_local), device=self.device)
        return z

    def noise_from_named(self, named, n):
        """{variable name: [N, 1, rows, cols]}  ->  [n_rows, N] (the latent vector's row order)"""
        out = np.zeros((self.program.n_rows, n), dtype=np.float32)
        for t in self.program.tensors:
            a = np.asarray(named[t["name"]], dtype=np.float32)
            out[t["row0"]:t["row0"] + t["size"]] = a.reshape(a.shape[0], -1).T
        return out

    def named_noise(self, noise, n):
        """[n_rows, N] -> {variable name: [N, 1, rows, cols]} (what the oracle takes)"""
        noise = np.asarray(noise)
        return {t["name"]: noise[t["row0"]:t["row0"] + t["size"]].T.reshape(n, 1, t["rows"], t["cols"]) for t in self.program.tensors}

    def _noise_tensor(self, noise, n_global, base, n_local):
        if noise is None:
            return None
        if isinstance(noise, dict):
            noise = self.noise_from_named(noise, n_global)
        if isinstance(noise, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(noise[:, base:base + n_local], dtype=np.float32)).to(self.device)
        return noise if noise.shape[1] == n_local else noise[:, base:base + n_local].contiguous()

    def _indices_tensor(self, minibatch):
        if minibatch is None:
            return None
        if isinstance(minibatch, dict):
            minibatch = minibatch[self.program.indices_name]
        idx = np.asarray(minibatch, dtype=np.int64).reshape(-1)
        # the kernel reads labels[row] and dataset[row] (csrc/bnn_kernel.inc, bnn_gather): rows outside the dataset are refused here
        if idx.size != self.program.batch_size or (idx.size and (idx.min() < 0 or idx.max() >= self.program.dataset_size)):
            raise ValueError("minibatch indices must be {} rows in [0, {}): got {} rows in [{}, {}]".format(
                self.program.batch_size, self.program.dataset_size, idx.size, idx.min() if idx.size else "-", idx.max() if idx.size else "-"))
        return torch.as_tensor(idx.astype(np.int32)).to(self.device)

    def evaluate(self, number_samples, noise=None, minibatch=None, seed=None, offset=None, want_noise=False,
                 want_fvalues=False, want_indices=False, **_):
        from brancher_amd import engine
        rank, world = engine.dist_info()
        base, n_local = engine.shard(number_samples, rank, world)
        if n_local == 0:
            raise ValueError("number_samples={} is smaller than the number of GPUs {}".format(number_samples, world))
        if offset is None:
            offset = self.iteration
            self.iteration += 1
        dev, p = self.device, self.program
        noise_t = self._noise_tensor(noise, number_samples, base, n_local)
        idx_t = self._indices_tensor(minibatch)
        noise_o = torch.empty((p.n_noise, n_local), device=dev) if want_noise else None
        idx_o = torch.empty(p.batch_size, device=dev, dtype=torch.int32) if want_indices else None
        fvals = torch.empty(n_local, device=dev) if want_fvalues else None
        logq = torch.zeros(n_local, device=dev) if want_fvalues and getattr(p, "estimator", "pathwise") == "blackbox" else None
        args = self._args(n_local, number_samples, base, noise_t, idx_t, seed, offset, noise_o, idx_o, fvals, logq)
        native.check(self.lib.bsvi_bnn_fwd_bwd(self.handle, C.byref(args)))
        engine.allreduce_sums(self.out)
        engine.check_exchange(self.device, self.params)
        native.check(self.lib.bsvi_bnn_finalize(self.handle, C.c_void_p(self.out.data_ptr()), number_samples, self._stream()))
        self.grads_valid = True
        res = dict(loss=self.out[2], finite=self.out[3], nonfinite_count=self.out[1],
                   grads=self.out[OUT_HEADER:OUT_HEADER + p.n_params], n_local=n_local, sample_base=base)
        if want_noise:
            res["noise"] = noise_o
        if want_indices:
            res["indices"] = idx_o
        if want_fvalues:
            res["f"] = fvals
            if logq is not None:
                res["lq"] = logq
        return res

    def evaluate_weighted(self, number_samples, f_weight, q_weight, seed, offset, noise=None, minibatch=None):
        """The second pass of a user-defined gradient estimator (`engine.custom_estimator_loss`), as `CompiledDense.evaluate_weighted`:
        the draw and the minibatch of (seed, offset) again, and -(sum_n a_n grad f_n + b_n grad log q_n) in the output block
        (bsvi_bnn_args::f_weight_dev / q_weight_dev; the model must have been created with the BlackBox estimator)."""
        from brancher_amd import engine
        rank, world = engine.dist_info()
        base, n_local = engine.shard(number_samples, rank, world)
        a = f_weight.reshape(-1)[base:base + n_local].contiguous().float()
        b = q_weight.reshape(-1)[base:base + n_local].contiguous().float()
        noise_t = self._noise_tensor(noise, number_samples, base, n_local)
        idx_t = self._indices_tensor(minibatch)
        args = self._args(n_local, number_samples, base, noise_t, idx_t, seed, int(offset), f_weight=a, q_weight=b)
        native.check(self.lib.bsvi_bnn_fwd_bwd(self.handle, C.byref(args)))
        engine.allreduce_sums(self.out)
        engine.check_exchange(self.device, self.params)
        native.check(self.lib.bsvi_bnn_finalize(self.handle, C.c_void_p(self.out.data_ptr()), 1, self._stream()))
        self.grads_valid = True
        return self.out[OUT_HEADER:OUT_HEADER + self.program.n_params]

    def _seed(self, seed):
        return _engine.shared_seed(seed, self.device)

    def named_grads(self):
        g = self.out[OUT_HEADER:].detach().cpu().numpy()
        return {par.name: g[off:off + size].reshape(par.shape).copy() for par, off, size, _ in self.program.parameters}

    def named_params(self):
        t = self.params.detach().cpu().numpy()
        return {par.name: t[off:off + size].reshape(par.shape).copy() for par, off, size, _ in self.program.parameters}

    def train(self, number_iterations, number_samples, optimizer="Adam", noise_seq=None, minibatch_seq=None, seed=None,
              pretraining_iterations=0, allow_persistent=True, _force_sharded_path=False, **opt_params):
        from brancher_amd import engine
        cfg = native.make_opt_cfg(optimizer, **opt_params)
        rank, world = engine.dist_info()
        base, n_local = engine.shard(number_samples, rank, world)
        if n_local == 0:
            raise ValueError("number_samples={} is smaller than the number of GPUs {}".format(number_samples, world))
        dev, p = self.device, self.program
        K = int(number_iterations)
        engine.broadcast_from_rank0(self.params)      # ranks step their own copies: they must start from the same values
        loss_curve, finite, state = engine.training_buffers(K, p.n_params, dev)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        offset0 = self.iteration
        self.iteration += K
        self.grads_valid = True
        # (the call's Philox key once: with seed=None on several ranks `shared_seed` is a broadcast and a host sync — not per iteration)
        seed = int(engine.shared_seed(seed, dev))
        for it in range(K):
            nz = None if noise_seq is None else self._noise_tensor(noise_seq[it], number_samples, base, n_local)
            mb = None if minibatch_seq is None else self._indices_tensor(minibatch_seq[it])
            args = self._args(n_local, number_samples, base, nz, mb, seed, offset0 + it)
            mask = self.mask_all if it > pretraining_iterations else self.mask_first
            if world == 1 and not _force_sharded_path:
                native.check(self.lib.bsvi_bnn_step(self.handle, C.byref(args), C.byref(cfg), ptr(self.params), ptr(state),
                                                    ptr(mask), C.c_void_p(loss_curve.data_ptr() + 4 * it),
                                                    C.c_void_p(finite.data_ptr() + 4 * it)))
            else:
                native.check(self.lib.bsvi_bnn_fwd_bwd(self.handle, C.byref(args)))
                engine.allreduce_sums(self.out)
                native.check(self.lib.bsvi_finalize_step(
                    C.byref(cfg), ptr(self.params), ptr(self.out), ptr(state), ptr(mask), p.n_params, number_samples,
                    C.c_void_p(loss_curve.data_ptr() + 4 * it), C.c_void_p(finite.data_ptr() + 4 * it), self._stream()))
        self.last_mode = "stepwise" if world == 1 else "stepwise+allreduce"
        if world > 1:
            engine.check_exchange(self.device, self.params)
        return loss_curve[:K], finite[:K]


# every native call of a compiled program runs with its device current (engine._bound_to_device)
from brancher_amd import engine as _engine  # noqa: E402  (engine imports this module lazily)
CompiledBnn = _engine._bound_to_device(CompiledBnn)
