"""
Dense-link models (BASELINE config 4): lowering and engine for graphs whose likelihood goes through
``BF.matmul(weights, x)`` with a random minibatch ``x``.

The scalar fused kernel keeps one latent *scalar* per LDS slot; a 10x784 weight matrix per sample
does not belong there.  This module recognises the reference's Bayesian logistic-regression shape
(`examples/MNIST_logistic_regression.py:15-54`, `examples/minibatch_logistic_regression.py:13-51`)

    indices = RandomIndices(dataset_size, batch_size, "indices", is_observed=True)
    x       = EmpiricalVariable(X,      indices=indices, name="x",      is_observed=True)
    labels  = EmpiricalVariable(labels, indices=indices, name="labels", is_observed=True)
    weights = NormalVariable(loc0, scale0, "weights")
    k       = CategoricalVariable(logits=BF.matmul(weights, x), name="k");  k.observe(labels)
    q:        NormalVariable(loc, scale, "weights", learnable=True)

and hands it to the MFMA kernels of `csrc/dense_kernel.inc` through `bsvi_dense_*` (include/bsvi.h).
The reference's semantics are kept, including the name-collision rule that makes the prior's
`weights_loc` / `weights_scale` the posterior's own roots and the absence of any
minibatch rescaling of the likelihood (`variables.py:849` TODO).
"""
import ctypes as C

import numpy as np
import torch

from brancher_amd import distributions as D
from brancher_amd import lowering, native
from brancher_amd.lowering import LoweringError, _Lowering
from brancher_amd.native import OUT_HEADER
from brancher_amd.variables import RandomVariable, RootVariable

LIK_CATEGORICAL, LIK_BERNOULLI = 0, 1


from brancher_amd.native import DenseDesc, DenseArgs


class DenseProgram:
    """What `lower_dense` extracts from the graph."""
    estimator = "pathwise"

    def summary(self):
        return dict(kind="dense", n_classes=self.n_classes, n_features=self.n_features,
                    dataset_size=self.dataset_size, batch_size=self.batch_size, n_params=self.n_params,
                    likelihood=("categorical", "bernoulli")[self.likelihood], latent=self.latent_name)


def _is_random(v):
    return isinstance(v, RandomVariable) and getattr(v, "_type", None) != "Deterministic node"


def lower_dense(joint, posterior, estimator="pathwise"):
    if estimator not in ("pathwise", "blackbox"):
        raise LoweringError("the dense-link path implements the Pathwise and BlackBox estimators")
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
    # point estimate (MAP, inference.py:251-275; examples/MAP_logistic_regression.py:46-56): the posterior is ONE
    # learnable RootVariable carrying the weight variable's name -> W = its value, no noise, no entropy
    point = None
    if not q_random:
        roots = [v for v in q_flat if isinstance(v, RootVariable) and v.learnable]
        if len(roots) == 1:
            point = roots[0]
    if point is None and (len(q_random) != 1 or q_random[0].distribution.kind != D.DIST_NORMAL):
        raise LoweringError("dense path: the posterior must be one mean-field Normal weight variable "
                            "(or one learnable RootVariable: a point estimate)")
    Wq = point if point is not None else q_random[0]
    p_random = [v for v in joint._flatten() if _is_random(v)]
    liks = [v for v in p_random if v.distribution.kind in (D.DIST_CATEGORICAL, D.DIST_BINOMIAL, D.DIST_BERNOULLI)]
    weights = [v for v in p_random if v.name == Wq.name and v.distribution.kind == D.DIST_NORMAL]
    others = [v for v in p_random if v not in liks and v not in weights and v.distribution.kind != D.DIST_EMPIRICAL]
    if len(liks) != 1 or len(weights) != 1 or others:
        raise LoweringError("dense path: expected one weight prior and one matmul likelihood")
    k, Wp = liks[0], weights[0]
    if not k.is_observed or not k.has_random_dataset:
        raise LoweringError("dense path: the likelihood must be observed through an EmpiricalVariable of labels")
    links = k.link.expressions()
    if "logits" not in links:
        raise LoweringError("dense path: the likelihood must be parameterised by logits")
    e = links["logits"].expr
    if not (e.op == "call" and e.attr[0] == "matmul" and len(e.args) == 2 and all(a.op == "var" for a in e.args)):
        raise LoweringError("dense path: logits must be BF.matmul(weights, x)")
    w_var, x_var = e.args[0].attr, e.args[1].attr
    if w_var is not Wp:
        raise LoweringError("dense path: the first matmul operand must be the weight variable")
    labels_var = k.dataset

    def minibatch_source(v, what):
        if getattr(v, "_type", None) != "Empirical" or not v.is_observed:
            raise LoweringError("dense path: %s must be an observed EmpiricalVariable" % what)
        exprs = v.link.expressions()
        ds = exprs["dataset"].expr
        if ds.op != "var" or not isinstance(ds.attr, RootVariable) or "indices" not in exprs:
            raise LoweringError("dense path: %s must index an array dataset through a RandomIndices variable" % what)
        ind = exprs["indices"].expr
        from brancher_amd.standard_variables import RandomIndices
        if ind.op != "var" or not isinstance(ind.attr, RandomIndices):
            raise LoweringError("dense path: %s must be indexed by a RandomIndices variable" % what)
        return np.asarray(ds.attr.value, dtype=np.float32), ind.attr

    X, ind_x = minibatch_source(x_var, "x")
    Y, ind_y = minibatch_source(labels_var, "labels")
    if ind_x is not ind_y:
        raise LoweringError("dense path: x and labels must share one RandomIndices variable")
    # observed datasets are stored as [1, DS, P, 1] / [1, DS, 1, 1] (utilities.py:226-232)
    DS = X.shape[1]
    Xm = X.reshape(DS, -1)
    Ym = Y.reshape(-1)
    if Ym.shape[0] != DS:
        raise LoweringError("dense path: dataset sizes of x and labels differ")
    P = Xm.shape[1]
    batch = int(ind_x.batch_size)

    # -- weight parameters through the uniform table
    def row_params(var, ctx):
        out = []
        for node in L.node_params(var, ctx):
            m = L.match_uniform(node)
            if m is None:
                raise LoweringError("dense path: the parameters of %r must be constants or parameter transforms" % var.name)
            leaf, g, a, b = m
            is_param, k0 = L.uniform_entries(leaf, g, a, b)
            size = int(np.prod(leaf.shape))
            out.append((is_param, k0, size, leaf.shape))
        return out

    if point is not None:
        leaf = L.q_value(point)
        is_param, k0 = L.uniform_entries(leaf, "identity", 0.0, 1.0)
        ql = (is_param, k0, int(np.prod(leaf.shape)), leaf.shape)
        _, z0 = L.const_operand(0.0)                         # scale 0: W = loc exactly, whatever the noise
        qs = (False, z0, 1, (1, 1, 1))
    else:
        (ql, qs) = row_params(Wq, L.q_value)
    (pl, ps) = row_params(Wp, L.p_value)
    shape = ql[3]
    C_, P_ = shape[1], shape[2]
    if shape[0] != 1 or P_ != P or ql[2] != C_ * P_:
        raise LoweringError("dense path: weights of shape %r do not match %d features" % (shape, P))
    lik = LIK_CATEGORICAL if k.distribution.kind == D.DIST_CATEGORICAL else LIK_BERNOULLI
    if lik == LIK_BERNOULLI:
        if C_ != 1:
            raise LoweringError("dense path: a Bernoulli/Binomial likelihood needs a single output")
        if k.distribution.kind == D.DIST_BINOMIAL:
            tc = L.match_uniform(L.from_expr(links["total_count"].expr, L.p_value))
            if tc is None or tc[0].op != "root" or float(np.asarray(tc[0].attr.value).reshape(-1)[0]) != 1.0:
                raise LoweringError("dense path: Binomial likelihood supports total_count = 1 only")

    uni, n_up = L.uniform_table()
    prog = DenseProgram()
    prog.estimator = estimator
    L.fill_parameter_tables(prog, uni, n_up)

    def base(entry):
        is_param, k0, size, _ = entry
        return (k0 if is_param else n_up + k0), (1 if size > 1 else 0)

    prog.uniform = uni
    prog.consts = np.concatenate(L.consts) if L.consts else np.zeros(0, np.float32)
    prog.q_loc, prog.q_scale, prog.prior_loc, prog.prior_scale = base(ql), base(qs), base(pl), base(ps)
    for b_, size in ((ql, 0), (qs, 0), (pl, 0), (ps, 0)):
        if b_[2] not in (1, C_ * P_):
            raise LoweringError("dense path: a weight parameter has an unsupported shape")
    prog.n_classes, prog.n_features, prog.dataset_size, prog.batch_size = C_, P_, DS, batch
    prog.likelihood = lik
    prog.dataset = np.ascontiguousarray(Xm, dtype=np.float32)
    prog.labels = np.ascontiguousarray(Ym, dtype=np.float32)
    prog.latent_name = Wq.name
    prog.point_estimate = point is not None
    prog.indices_name = ind_x.name
    prog.lik_weight, prog.prior_weight, prog.entropy_weight = 1.0, 1.0, (0.0 if point is not None else 1.0)
    prog.n_noise = C_ * P_
    prog.bmax = 1
    return prog


class CompiledDense:
    """Engine for a dense-link model; same surface as engine.CompiledELBO."""

    def data_path(self):
        """which matrix-core path serves the two products of an iteration: "bf16x3" when every dataset value is exactly a
        bf16 number (three bf16 MFMAs on the exact pieces of the f32 operand, bsvi_dense_exact_data), else "f32" """
        return "bf16x3" if self._exact_data else "f32"

    def __init__(self, joint_model, posterior_model, estimator="pathwise", device=None, program=None):
        from brancher_amd import engine
        self.device = device or engine._device()
        self.program = program if program is not None else lower_dense(joint_model, posterior_model, estimator)
        p = self.program
        lib = native.load()
        if lib.bsvi_device_count() < 1:
            raise native.NativeError("no MI355X / HIP device visible: the engine cannot run (no CPU fallback)")
        self.lib = lib
        self._keep = dict(uniform=np.ascontiguousarray(p.uniform), consts=np.ascontiguousarray(p.consts, dtype=np.float32),
                          ptr=np.ascontiguousarray(p.param_uniform_ptr, dtype=np.uint32),
                          idx=np.ascontiguousarray(p.param_uniform_idx, dtype=np.uint32),
                          dataset=p.dataset, labels=p.labels)
        k = self._keep
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None
        d = DenseDesc(abi_version=native.ABI_VERSION, n_params=p.n_params, n_consts=k["consts"].size,
                      n_uniform=len(k["uniform"]), n_uniform_grad=p.n_uniform_grad,
                      n_classes=p.n_classes, n_features=p.n_features, dataset_size=p.dataset_size,
                      batch_size=p.batch_size, likelihood=p.likelihood,
                      q_loc_u=p.q_loc[0], q_scale_u=p.q_scale[0], prior_loc_u=p.prior_loc[0], prior_scale_u=p.prior_scale[0],
                      q_loc_stride=p.q_loc[1], q_scale_stride=p.q_scale[1], prior_loc_stride=p.prior_loc[1],
                      prior_scale_stride=p.prior_scale[1], lik_weight=p.lik_weight, prior_weight=p.prior_weight,
                      entropy_weight=p.entropy_weight, estimator=lowering.EST[getattr(p, "estimator", "pathwise")],
                      uniform=ptr(k["uniform"]), consts=ptr(k["consts"]),
                      param_uniform_ptr=ptr(k["ptr"]), param_uniform_idx=ptr(k["idx"]), dataset=ptr(k["dataset"]),
                      labels=ptr(k["labels"]))
        handle = C.c_void_p()
        native.check(lib.bsvi_dense_create(C.byref(d), C.byref(handle)))
        self.handle = handle
        self._exact_data = bool(lib.bsvi_dense_exact_data(handle))
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
                self.lib.bsvi_dense_destroy(self.handle)
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
            ws = torch.empty(int(self.lib.bsvi_dense_workspace_bytes(self.handle, n_local)), dtype=torch.uint8,
                             device=self.device)
            self._workspaces[n_local] = ws
        return ws

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _seed(self, seed):
        return _engine.shared_seed(seed, self.device)

    def _args(self, n_local, n_global, base, noise=None, indices=None, seed=None, offset=0, noise_out=None,
              indices_out=None, fvalue_out=None, f_weight=None, q_weight=None, logq_out=None):
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        seed = _engine.shared_seed(seed, self.device)
        return DenseArgs(params_dev=ptr(self.params), noise_dev=ptr(noise), indices_dev=ptr(indices), seed=seed,
                         offset=int(offset), n_samples_local=n_local, n_samples_global=n_global, sample_base=base,
                         out_dev=ptr(self.out), noise_out_dev=ptr(noise_out), indices_out_dev=ptr(indices_out),
                         fvalue_out_dev=ptr(fvalue_out), workspace_dev=ptr(self.workspace(n_local)), stream=self._stream(),
                         f_weight_dev=ptr(f_weight), q_weight_dev=ptr(q_weight), logq_out_dev=ptr(logq_out))

    def _noise_tensor(self, noise, n_global, base, n_local):
        if noise is None:
            return None
        if isinstance(noise, dict):
            if self.program.point_estimate and self.program.latent_name not in noise:
                return None                  # nothing is sampled: the weights' scale is 0, any noise gives W = loc
            a = np.asarray(noise[self.program.latent_name], dtype=np.float32)       # [N, 1, C, P]
            noise = np.ascontiguousarray(a.reshape(a.shape[0], -1).T)               # [C*P, N]
        if isinstance(noise, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(noise[:, base:base + n_local], dtype=np.float32)).to(self.device)
        return noise if noise.shape[1] == n_local else noise[:, base:base + n_local].contiguous()

    def _indices_tensor(self, minibatch):
        if minibatch is None:
            return None
        if isinstance(minibatch, dict):
            minibatch = minibatch[self.program.indices_name]
        return torch.as_tensor(np.asarray(minibatch, dtype=np.int32)).to(self.device)

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
        # (per-sample log q next to per-sample f: what a BlackBox-style estimator multiplies — a point estimate has no q)
        logq = torch.zeros(n_local, device=dev) if want_fvalues and getattr(p, "estimator", "pathwise") == "blackbox" else None
        args = self._args(n_local, number_samples, base, noise_t, idx_t, seed, offset, noise_o, idx_o, fvals, logq_out=logq)
        native.check(self.lib.bsvi_dense_fwd_bwd(self.handle, C.byref(args)))
        engine.allreduce_sums(self.out)
        engine.check_exchange(self.device, self.params)
        native.check(self.lib.bsvi_dense_finalize(self.handle, C.c_void_p(self.out.data_ptr()), number_samples, self._stream()))
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
        """The second pass of a user-defined gradient estimator (`engine.custom_estimator_loss`), as
        `CompiledELBO.evaluate_weighted`: the draw and the minibatch of (seed, offset) again, and
        -(sum_n a_n grad f_n + b_n grad log q_n) in the output block (bsvi_dense_args::f_weight_dev / q_weight_dev; the
        model must have been created with the BlackBox estimator)."""
        from brancher_amd import engine
        rank, world = engine.dist_info()
        base, n_local = engine.shard(number_samples, rank, world)
        a = f_weight.reshape(-1)[base:base + n_local].contiguous().float()
        b = q_weight.reshape(-1)[base:base + n_local].contiguous().float()
        noise_t = self._noise_tensor(noise, number_samples, base, n_local)
        idx_t = self._indices_tensor(minibatch)
        args = self._args(n_local, number_samples, base, noise_t, idx_t, seed, int(offset), f_weight=a, q_weight=b)
        native.check(self.lib.bsvi_dense_fwd_bwd(self.handle, C.byref(args)))
        engine.allreduce_sums(self.out)
        engine.check_exchange(self.device, self.params)
        native.check(self.lib.bsvi_dense_finalize(self.handle, C.c_void_p(self.out.data_ptr()), 1, self._stream()))
        self.grads_valid = True
        return self.out[OUT_HEADER:OUT_HEADER + self.program.n_params]

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
        for it in range(K):
            nz = None if noise_seq is None else self._noise_tensor(noise_seq[it], number_samples, base, n_local)
            mb = None if minibatch_seq is None else self._indices_tensor(minibatch_seq[it])
            args = self._args(n_local, number_samples, base, nz, mb, seed, offset0 + it)
            mask = self.mask_all if it > pretraining_iterations else self.mask_first
            if world == 1 and not _force_sharded_path:
                native.check(self.lib.bsvi_dense_step(self.handle, C.byref(args), C.byref(cfg), ptr(self.params), ptr(state),
                                                      ptr(mask), C.c_void_p(loss_curve.data_ptr() + 4 * it),
                                                      C.c_void_p(finite.data_ptr() + 4 * it)))
            else:
                native.check(self.lib.bsvi_dense_fwd_bwd(self.handle, C.byref(args)))
                engine.allreduce_sums(self.out)
                native.check(self.lib.bsvi_finalize_step(
                    C.byref(cfg), ptr(self.params), ptr(self.out), ptr(state), ptr(mask), p.n_params, number_samples,
                    C.c_void_p(loss_curve.data_ptr() + 4 * it), C.c_void_p(finite.data_ptr() + 4 * it), self._stream()))
        self.last_mode = "stepwise" if world == 1 else "stepwise+allreduce"
        if world > 1:
            engine.check_exchange(self.device, self.params)         # (an abandoned exchange poisoned a step: say so, loudly)
        return loss_curve[:K], finite[:K]


# every native call of a compiled program runs with its device current (engine._bound_to_device)
from brancher_amd import engine as _engine  # noqa: E402  (engine imports this module lazily)
CompiledDense = _engine._bound_to_device(CompiledDense)
