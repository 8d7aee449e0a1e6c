"""
Amortised models (BASELINE config 5): lowering and engine for graphs whose posterior is computed by an encoder
network and whose likelihood goes through a decoder network (`examples/VAE_playground.py:18-88`):

    encoder = BF.BrancherFunction(EncoderModule)        # torch.nn.Module returning {"mean": ..., "sd": ...}
    decoder = BF.BrancherFunction(DecoderModule)        # torch.nn.Module returning {"mean": ...}
    z   = NormalVariable(np.zeros(Dz), np.ones(Dz), "z")
    out = DeterministicVariable(decoder(z), "decoder_output")
    x   = BinomialVariable(total_count=1, logits=out["mean"], name="x")
    q:    Qx  = EmpiricalVariable(dataset, batch_size=B, name="x", is_observed=True)
          enc = DeterministicVariable(encoder(Qx), "encoder_output")
          Qz  = NormalVariable(enc["mean"], enc["sd"], "z")

The reference runs the two modules through autograd on the N*B rows of an iteration (every Monte-Carlo sample
draws its own minibatch, `distributions.py:436-441`).  Here `trace_network` reads each module's structure once
with torch.fx — Linear layers, elementwise ReLU / Softplus, `+ constant`, dict outputs — and the iteration runs
on the f32 MFMA GEMMs of `csrc/amort_kernel.hip` through `bsvi_amort_*` (include/bsvi.h).  The modules' tensors
live in the engine's flat parameter buffer in torch's layout; `sync_modules()` writes them back.
"""
import ctypes as C
import operator
import os

import numpy as np
import torch

from brancher_amd import distributions as D
from brancher_amd import native
from brancher_amd.functions import ModuleLink
from brancher_amd.lowering import LoweringError
from brancher_amd.native import OUT_HEADER, AmortArgs, AmortDesc, MlpLayer
from brancher_amd.variables import RandomVariable, RootVariable

ACT_NONE, ACT_RELU, ACT_SOFTPLUS = 0, 1, 2
NO_BIAS = 0xFFFFFFFF


class Layer:
    __slots__ = ("in_value", "out_value", "n_in", "n_out", "weight", "bias", "activation", "post_add",
                 "weight_off", "bias_off", "split_col", "activation2", "post_add2", "parts")

    def __repr__(self):
        return "Layer(%d -> %d, %dx%d, act=%d, +%g)" % (self.in_value, self.out_value, self.n_out, self.n_in,
                                                        self.activation, self.post_add)


def trace_network(link):
    """Layer list of a ModuleLink: ([Layer...], {output key: value id}); value 0 is the module's input.

    Accepted graph (anything else raises LoweringError): `nn.Linear`; `nn.ReLU` / `relu`; `nn.Softplus()` /
    `softplus` with default beta and threshold; `value + python number`; `squeeze` / `flatten` of the INPUT
    (the reference stores a row as [P, 1]); a tensor or a dict of tensors as output.  An activation or constant
    shift is folded into the Linear layer that produces its operand, which therefore must have no other reader."""
    import torch.fx as fx
    import torch.nn as nn
    import torch.nn.functional as F
    module = link.module

    class _Root(nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x):
            return self.m(x)

    root = _Root(module)
    try:
        graph = fx.Tracer().trace(root)
    except Exception as e:  # noqa: BLE001 - fx raises many types
        raise LoweringError("amortised path: torch.fx could not trace %s: %s" % (type(module).__name__, e))
    layers, value_of, outputs = [], {}, {}
    produced_by = {}

    def fold(node, src, activation=None, post_add=None):
        v = value_of.get(src)
        layer = produced_by.get(v)
        if layer is None:
            raise LoweringError("amortised path: %s must follow a Linear layer" % node.name)
        if len(src.users) != 1:
            raise LoweringError("amortised path: the operand of %s is read elsewhere too" % node.name)
        if activation is not None:
            if layer.activation != ACT_NONE or layer.post_add != 0.0:
                raise LoweringError("amortised path: two activations on one layer (%s)" % node.name)
            layer.activation = activation
        if post_add is not None:
            layer.post_add += float(post_add)
        value_of[node] = v

    for node in graph.nodes:
        if node.op == "placeholder":
            if value_of:
                raise LoweringError("amortised path: a network link takes one input")
            value_of[node] = 0
        elif node.op == "call_method" and node.target in ("squeeze", "flatten", "float", "contiguous"):
            if value_of.get(node.args[0]) != 0:
                raise LoweringError("amortised path: %s is supported on the network input only" % node.target)
            value_of[node] = 0
        elif node.op == "call_module":
            sub = root.get_submodule(node.target)
            src = node.args[0]
            if isinstance(sub, nn.Linear):
                if src not in value_of:
                    raise LoweringError("amortised path: unsupported operand of %s" % node.target)
                lay = Layer()
                lay.in_value, lay.out_value = value_of[src], len(layers) + 1
                lay.n_in, lay.n_out = sub.in_features, sub.out_features
                prefix = node.target[2:] if node.target.startswith("m.") else node.target
                lay.weight = link.named[prefix + ".weight"]
                lay.bias = link.named[prefix + ".bias"] if sub.bias is not None else None
                lay.activation, lay.post_add = ACT_NONE, 0.0
                lay.split_col, lay.activation2, lay.post_add2, lay.parts = 0, ACT_NONE, 0.0, None
                layers.append(lay)
                value_of[node] = lay.out_value
                produced_by[lay.out_value] = lay
            elif isinstance(sub, nn.ReLU):
                fold(node, src, activation=ACT_RELU)
            elif isinstance(sub, nn.Softplus):
                if sub.beta != 1 or sub.threshold != 20:
                    raise LoweringError("amortised path: Softplus with non-default beta/threshold")
                fold(node, src, activation=ACT_SOFTPLUS)
            elif isinstance(sub, (nn.Identity, nn.Flatten)) and value_of.get(src) == 0:
                value_of[node] = 0
            else:
                raise LoweringError("amortised path: module %s is not supported" % type(sub).__name__)
        elif node.op == "call_function":
            if node.target in (F.relu, torch.relu):
                fold(node, node.args[0], activation=ACT_RELU)
            elif node.target is F.softplus and len(node.args) == 1 and not node.kwargs:
                fold(node, node.args[0], activation=ACT_SOFTPLUS)
            elif node.target in (operator.add, torch.add) and len(node.args) == 2:
                a, b = node.args
                if isinstance(b, fx.Node) and not isinstance(a, fx.Node):
                    a, b = b, a
                if not isinstance(a, fx.Node) or not isinstance(b, (int, float)):
                    raise LoweringError("amortised path: only `value + number` additions are supported")
                fold(node, a, post_add=b)
            else:
                raise LoweringError("amortised path: function %s is not supported" % getattr(node.target, "__name__", node.target))
        elif node.op == "output":
            res = node.args[0]
            if isinstance(res, dict):
                outputs = {k: value_of[v] for k, v in res.items()}
            elif isinstance(res, fx.Node):
                outputs = {None: value_of[res]}
            else:
                raise LoweringError("amortised path: a network must return a tensor or a dict of tensors")
        else:
            raise LoweringError("amortised path: unsupported graph node %s" % node.op)
    return layers, outputs


def merge_sibling_heads(layers, outputs):
    """Two narrow layers that read the same value and feed nothing further — the latent's loc and scale heads — become
    ONE layer: their weight rows (and biases) are laid out adjacently in the parameter buffer, the output columns of
    the second keep their own activation (`bsvi_mlp_layer.split_col`).  One pass over the trunk value instead of two,
    forward and backward.  Returns (layers, {key: (value, first column)}, [(weights...), (biases...)] adjacency groups)."""
    consumed = {l.in_value for l in layers}
    cols = {k: (v, 0) for k, v in outputs.items()}
    groups = []
    heads = [l for l in layers if l.out_value not in consumed and l.n_out <= 4]
    by_input = {}
    for l in heads:
        by_input.setdefault(l.in_value, []).append(l)
    for sibs in by_input.values():
        if len(sibs) != 2 or (sibs[0].bias is None) != (sibs[1].bias is None):
            continue
        first, second = sibs
        merged = Layer()
        merged.in_value, merged.out_value = first.in_value, first.out_value
        merged.n_in, merged.n_out = first.n_in, first.n_out + second.n_out
        merged.weight, merged.bias = first.weight, first.bias
        merged.activation, merged.post_add = first.activation, first.post_add
        merged.split_col, merged.activation2, merged.post_add2 = first.n_out, second.activation, second.post_add
        merged.parts = (first, second)
        groups.append((first.weight, second.weight))
        if first.bias is not None:
            groups.append((first.bias, second.bias))
        layers = [merged if l is first else l for l in layers if l is not second]
        for k, (v, c) in list(cols.items()):
            if v == second.out_value:
                cols[k] = (first.out_value, first.n_out)
    # value ids stay as they are (gaps are fine: the C side sizes its tables by the largest id)
    return layers, cols, groups


class AmortizedProgram:
    def summary(self):
        return dict(kind="amortized", n_params=self.n_params, n_features=self.n_features, latent_dim=self.latent_dim,
                    dataset_size=self.dataset_size, batch_size=self.batch_size, estimator=self.estimator,
                    encoder=[(l.n_in, l.n_out) for l in self.enc_layers],
                    decoder=[(l.n_in, l.n_out) for l in self.dec_layers])


def _is_random(v):
    return isinstance(v, RandomVariable) and getattr(v, "_type", None) != "Deterministic node"


def _network_output(expr, what):
    """(ModuleLink, input Variable, key) of `network(var)[key]`, looking through DeterministicVariables."""
    key = None
    e = expr
    if e.op == "getitem" and isinstance(e.attr, str):
        key, e = e.attr, e.args[0]
    if e.op == "var" and getattr(e.attr, "_type", None) == "Deterministic node":
        e = e.attr.link.expressions()["value"].expr
        if e.op == "getitem" and isinstance(e.attr, str) and key is None:
            key, e = e.attr, e.args[0]
    if not (e.op == "call" and isinstance(e.attr[0], ModuleLink) and len(e.args) == 1 and e.args[0].op == "var"):
        raise LoweringError("amortised path: %s must be the output of a network link applied to one variable" % what)
    return e.attr[0], e.args[0].attr, key


def _root_parameter(var, name, positive, learnable_ok=False):
    """(values, Parameter or None) of a distribution argument given as a number / array: the standard constructors store it
    as a RootVariable behind the range's forward transform (`standard_variables.py:57-68`, `geometric_ranges.py`): root, or
    0 + softplus(root).  With `learnable=True` on the constructor the root is a parameter; the values are its current ones."""
    link = var.link.expressions()[name]
    e = link.expr
    roots = [v for v in e.variables() if isinstance(v, RootVariable)]
    if len(roots) != 1 or len(e.variables()) != 1 or (roots[0].learnable and not learnable_ok):
        raise LoweringError("amortised path: %s of %r must be a constant" % (name, var.name))
    raw = np.asarray(roots[0].value, dtype=np.float64).reshape(-1)
    par = roots[0].parameter if roots[0].learnable else None
    if not positive:
        if e.op != "var":
            raise LoweringError("amortised path: unexpected transform of %s of %r" % (name, var.name))
        return raw, par
    return np.where(raw > 20, raw, np.log1p(np.exp(np.minimum(raw, 20)))), par


def _constant_parameter(var, name, positive):
    return _root_parameter(var, name, positive)[0]


def lower_amortized(joint, posterior, estimator="pathwise"):
    if estimator not in ("pathwise", "blackbox"):
        raise LoweringError("the amortised path implements the Pathwise and BlackBox estimators")
    q_flat, p_flat = posterior._flatten(), joint._flatten()
    q_random = [v for v in q_flat if _is_random(v)]
    q_emp = [v for v in q_random if v.distribution.kind == D.DIST_EMPIRICAL]
    q_lat = [v for v in q_random if v.distribution.kind != D.DIST_EMPIRICAL]
    if len(q_emp) != 1 or len(q_lat) != 1 or q_lat[0].distribution.kind != D.DIST_NORMAL:
        raise LoweringError("amortised path: the posterior must be one EmpiricalVariable and one Normal latent")
    Qx, Qz = q_emp[0], q_lat[0]
    if not Qx.is_observed:
        raise LoweringError("amortised path: the minibatch variable must be observed")
    ex = Qx.link.expressions()
    if "indices" in ex or "weights" in ex:
        raise LoweringError("amortised path: the minibatch variable draws its own rows (batch_size=...)")
    ds = ex["dataset"].expr
    if ds.op != "var" or not isinstance(ds.attr, RootVariable):
        raise LoweringError("amortised path: the dataset must be an array")
    X = np.asarray(ds.attr.value, dtype=np.float32)          # observed datasets: [1, DS, P, 1] (utilities.py:226-232)
    DS = X.shape[1]
    Xm = np.ascontiguousarray(X.reshape(DS, -1))
    P, B = Xm.shape[1], int(Qx.batch_size)

    ql = Qz.link.expressions()
    enc_link, enc_in, loc_key = _network_output(ql["loc"].expr, "the latent's loc")
    enc_link2, enc_in2, scale_key = _network_output(ql["scale"].expr, "the latent's scale")
    if enc_link is not enc_link2 or enc_in is not Qx or enc_in2 is not Qx:
        raise LoweringError("amortised path: loc and scale must be outputs of ONE encoder applied to the minibatch")

    p_random = [v for v in p_flat if _is_random(v)]
    lik = [v for v in p_random if v.name == Qx.name]
    lat = [v for v in p_random if v.name == Qz.name]
    if len(lik) != 1 or len(lat) != 1 or len(p_random) != 2:
        raise LoweringError("amortised path: the model must be one latent prior and one likelihood named like the "
                            "posterior's variables")
    x, z = lik[0], lat[0]
    if z.distribution.kind != D.DIST_NORMAL:
        raise LoweringError("amortised path: the latent prior must be Normal")
    if x.distribution.kind not in (D.DIST_BINOMIAL, D.DIST_BERNOULLI, D.DIST_NORMAL):
        raise LoweringError("amortised path: the likelihood must be Binomial(1, logits) / Bernulli(logits) or "
                            "Normal(decoder output, scale given as numbers — constant or learnable)")
    xl = x.link.expressions()
    likelihood, lik_scale, lik_scale_par, scale_head_key = "binomial", None, None, None
    if x.distribution.kind == D.DIST_NORMAL:
        likelihood = "normal"
        dec_link, dec_in, logits_key = _network_output(xl["loc"].expr, "the likelihood's loc")
        try:
            # a second HEAD of the decoder: NormalVariable(decoder(z)["mean"], decoder(z)["sd"])
            sd_link, sd_in, scale_head_key = _network_output(xl["scale"].expr, "the likelihood's scale")
            if sd_link is not dec_link or sd_in is not dec_in or scale_head_key is None or scale_head_key == logits_key:
                raise LoweringError("amortised path: the likelihood's loc and scale must be two outputs of ONE decoder")
            lik_scale = np.ones(P, dtype=np.float32)          # (unused by the kernels: the head's values are)
        except LoweringError as head_error:
            if scale_head_key is not None:
                raise head_error
            # a number / array: a constant, or — `NormalVariable(decoder value, scale, learnable=True)` — a parameter of the joint
            # model behind softplus (standard_variables.py:57-68)
            lik_scale, lik_scale_par = _root_parameter(x, "scale", positive=True, learnable_ok=True)
            if lik_scale.size not in (1, P):
                raise LoweringError("amortised path: the likelihood's scale must be one number or one per feature")
            lik_scale = np.ascontiguousarray(np.broadcast_to(lik_scale, (P,)), dtype=np.float32)
    else:
        if "logits" not in xl:
            raise LoweringError("amortised path: the likelihood must be parameterised by logits")
        if x.distribution.kind == D.DIST_BINOMIAL:
            tc = _constant_parameter(x, "total_count", positive=False)
            if tc.size != 1 or tc[0] != 1.0:
                raise LoweringError("amortised path: Binomial likelihood supports total_count = 1 only")
        dec_link, dec_in, logits_key = _network_output(xl["logits"].expr, "the likelihood's logits")
    if dec_in is not z:
        raise LoweringError("amortised path: the decoder must be applied to the latent variable")
    # the prior's loc / scale: constants, or (NormalVariable(..., learnable=True)) parameters of the joint model
    prior_loc, prior_loc_par = _root_parameter(z, "loc", positive=False, learnable_ok=True)
    prior_scale, prior_scale_par = _root_parameter(z, "scale", positive=True, learnable_ok=True)
    Dz = prior_loc.size
    if prior_scale.size not in (1, Dz) or (prior_scale_par is not None and prior_scale.size != Dz):
        raise LoweringError("amortised path: prior loc / scale shapes differ")
    prior_scale = np.broadcast_to(prior_scale, (Dz,))

    enc_layers, enc_out = trace_network(enc_link)
    dec_layers, dec_out = trace_network(dec_link)
    for key, outs, what in ((loc_key, enc_out, "encoder"), (scale_key, enc_out, "encoder"), (logits_key, dec_out, "decoder")) + \
            (((scale_head_key, dec_out, "decoder"),) if scale_head_key is not None else ()):
        if key not in outs:
            raise LoweringError("amortised path: the %s has no output %r" % (what, key))
    enc_layers, enc_cols, adjacent = merge_sibling_heads(enc_layers, enc_out)

    prog = AmortizedProgram()
    prog.estimator = estimator
    # parameter buffer: group 0 = the posterior's optimizer (encoder), group 1 = the joint model's (decoder)
    # (inference.py:77-88); weights first so that every matrix starts 16-byte aligned when its size allows
    prog.parameters, off = [], 0
    offsets = {}
    for group, link in ((0, enc_link), (1, dec_link)):
        off = (off + 3) // 4 * 4          # every network starts 16-byte aligned (padding elements stay inactive)
        pars = sorted(link.parameters(), key=lambda p: (p.size % 4 != 0, len(p.shape) < 2))
        # merged sibling layers need their tensors back to back, in layer order
        for grp in adjacent:
            if all(any(q is g for q in pars) for g in grp):
                at = min(i for i, q in enumerate(pars) if any(q is g for g in grp))
                rest = [q for q in pars if not any(q is g for g in grp)]
                pars = rest[:at] + list(grp) + rest[at:]
        for par in pars:
            if id(par) in offsets:
                continue
            offsets[id(par)] = off
            prog.parameters.append((par, off, par.size, group))
            off += par.size
    # a learnable prior belongs to the joint model's optimizer (inference.py:77-88: ProbabilisticOptimizer(joint_model))
    prog.prior_loc_off = prog.prior_scale_off = NO_BIAS
    for par, which in ((prior_loc_par, "prior_loc_off"), (prior_scale_par, "prior_scale_off")):
        if par is not None:
            setattr(prog, which, off)
            prog.parameters.append((par, off, par.size, 1))
            off += par.size
    prog.lik_scale_off, prog.lik_scale_size = NO_BIAS, 0
    if lik_scale_par is not None:
        prog.lik_scale_off, prog.lik_scale_size = off, lik_scale_par.size
        prog.parameters.append((lik_scale_par, off, lik_scale_par.size, 1))
        off += lik_scale_par.size
    prog.n_params = off
    prog.param_active = np.zeros(off, dtype=np.uint8)
    for par, o, size, g in prog.parameters:
        prog.param_active[o:o + size] = 1
    group = np.zeros(off, dtype=np.uint8)
    for par, o, size, g in prog.parameters:
        group[o:o + size] = g
    prog.param_group = group
    for lay in enc_layers + dec_layers:
        lay.weight_off = offsets[id(lay.weight)]
        lay.bias_off = offsets[id(lay.bias)] if lay.bias is not None else NO_BIAS
    if enc_layers[0].n_in != P and any(l.in_value == 0 and l.n_in != P for l in enc_layers):
        raise LoweringError("amortised path: the encoder's input width does not match the dataset rows")
    if any(l.in_value == 0 and l.n_in != Dz for l in dec_layers):
        raise LoweringError("amortised path: the decoder's input width does not match the latent")
    prog.enc_layers, prog.dec_layers = enc_layers, dec_layers
    prog.enc_outputs, prog.dec_outputs, prog.logits_key = enc_cols, {k: (v, 0) for k, v in dec_out.items()}, logits_key
    (prog.enc_loc_value, prog.enc_loc_col), (prog.enc_scale_value, prog.enc_scale_col) = enc_cols[loc_key], enc_cols[scale_key]
    prog.dec_logits_value = dec_out[logits_key]
    prog.dec_scale_key = scale_head_key
    prog.dec_scale_value = dec_out[scale_head_key] if scale_head_key is not None else 0
    prog.n_features, prog.latent_dim, prog.dataset_size, prog.batch_size = P, Dz, DS, B
    prog.prior_loc = np.ascontiguousarray(prior_loc, dtype=np.float32)
    prog.prior_scale = np.ascontiguousarray(prior_scale, dtype=np.float32)
    prog.dataset = Xm
    prog.likelihood, prog.likelihood_scale = likelihood, lik_scale
    prog.loc_key, prog.scale_key = loc_key, scale_key
    prog.latent_name, prog.data_name = Qz.name, Qx.name
    prog.links = (enc_link, dec_link)
    prog.n_noise = Dz
    return prog


class CompiledAmortized:
    """Engine for an amortised model; same surface as engine.CompiledELBO / dense.CompiledDense."""

    def data_path(self):
        """which matrix-core path serves the layers that read the data rows: "bf16x3" when every dataset value is exactly a
        bf16 number (three bf16 MFMAs on the exact pieces of the f32 weights), else "f32" (bsvi_amort_exact_data)"""
        return "bf16x3" if self._exact_data else "f32"

    def __init__(self, joint_model, posterior_model, estimator="pathwise", device=None, program=None):
        from brancher_amd import engine
        self.device = device or engine._device()
        self.program = p = program if program is not None else lower_amortized(joint_model, posterior_model, estimator)
        lib = native.load()
        if lib.bsvi_device_count() < 1:
            raise native.NativeError("no MI355X / HIP device visible: the engine cannot run (no CPU fallback)")
        self.lib = lib

        def pack(layers):
            arr = (MlpLayer * len(layers))()
            for k, l in enumerate(layers):
                arr[k] = MlpLayer(l.in_value, l.out_value, l.n_in, l.n_out, l.weight_off, l.bias_off, l.activation,
                                  l.post_add, l.split_col, l.activation2, l.post_add2, 0)
            return arr

        self._keep = dict(enc=pack(p.enc_layers), dec=pack(p.dec_layers), loc=p.prior_loc, scale=p.prior_scale,
                          dataset=p.dataset, lik_scale=p.likelihood_scale)
        k = self._keep
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        d = AmortDesc(abi_version=native.ABI_VERSION, n_params=p.n_params, n_features=p.n_features,
                      latent_dim=p.latent_dim, dataset_size=p.dataset_size, batch_size=p.batch_size,
                      n_enc_layers=len(p.enc_layers), n_dec_layers=len(p.dec_layers),
                      enc_loc_value=p.enc_loc_value, enc_scale_value=p.enc_scale_value,
                      enc_loc_col=p.enc_loc_col, enc_scale_col=p.enc_scale_col,
                      dec_logits_value=p.dec_logits_value, enc_layers=k["enc"], dec_layers=k["dec"],
                      prior_loc=ptr(k["loc"]), prior_scale=ptr(k["scale"]), dataset=ptr(k["dataset"]),
                      likelihood=1 if p.likelihood == "normal" else 0, likelihood_scale=ptr(k["lik_scale"]),
                      prior_loc_off=p.prior_loc_off, prior_scale_off=p.prior_scale_off,
                      lik_scale_off=p.lik_scale_off, lik_scale_size=p.lik_scale_size, dec_scale_value=p.dec_scale_value)
        handle = C.c_void_p()
        native.check(lib.bsvi_amort_create(C.byref(d), C.byref(handle)))
        self.handle = handle
        self._exact_data = bool(lib.bsvi_amort_exact_data(handle))
        dev = self.device
        self.n_params = p.n_params
        theta = np.zeros(p.n_params, dtype=np.float32)
        for par, off, size, _ in p.parameters:
            theta[off:off + size] = par.numpy().reshape(-1)
        self.params = _engine.broadcast_from_rank0(torch.from_numpy(theta).to(dev))
        self.out = torch.zeros(OUT_HEADER + max(p.n_params, 1), device=dev)
        active, group = p.param_active, p.param_group
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
                self.lib.bsvi_amort_destroy(self.handle)
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

    def sync_modules(self):
        """write the trained tensors back into the user's torch modules"""
        for link in self.program.links:
            link.sync_to_module()

    def workspace(self, n_local):
        ws = self._workspaces.get(n_local)
        if ws is None:
            ws = torch.empty(int(self.lib.bsvi_amort_workspace_bytes(self.handle, n_local)), dtype=torch.uint8,
                             device=self.device)
            self._workspaces[n_local] = ws
        return ws

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _noise_tensor(self, noise, base, n_local):
        """eps as [n_local * B, Dz]; accepts the reference's sample layout [N, B, Dz] keyed by the latent's name"""
        if noise is None:
            return None
        p = self.program
        if isinstance(noise, dict):
            noise = noise[p.latent_name]
        if isinstance(noise, np.ndarray):
            if noise.size % (p.batch_size * p.latent_dim):
                raise ValueError("noise must have shape [number_samples, %d, %d]" % (p.batch_size, p.latent_dim))
            a = np.asarray(noise, dtype=np.float32).reshape(-1, p.batch_size, p.latent_dim)[base:base + n_local]
            return torch.from_numpy(np.ascontiguousarray(a.reshape(-1, p.latent_dim))).to(self.device)
        return noise.reshape(-1, p.batch_size, p.latent_dim)[base:base + n_local].reshape(-1, p.latent_dim).contiguous()

    def _indices_tensor(self, minibatch, base, n_local):
        if minibatch is None:
            return None
        p = self.program
        if isinstance(minibatch, dict):
            minibatch = minibatch[p.data_name]
        a = np.asarray(minibatch, dtype=np.int64).reshape(-1, p.batch_size)[base:base + n_local]
        if a.size and (a.min() < 0 or a.max() >= p.dataset_size):
            raise ValueError("minibatch rows must lie in [0, %d)" % p.dataset_size)
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)

    def _seed(self, seed):
        return _engine.shared_seed(seed, self.device)

    def _args(self, n_local, n_global, base, noise=None, indices=None, seed=None, offset=0, noise_out=None,
              indices_out=None, fvalue_out=None, logq_out=None, f_weight=None, q_weight=None):
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        seed = _engine.shared_seed(seed, self.device)
        return AmortArgs(params_dev=ptr(self.params), noise_dev=ptr(noise), indices_dev=ptr(indices), seed=seed,
                         offset=int(offset), n_samples_local=n_local, n_samples_global=n_global, sample_base=base,
                         estimator=1 if self.program.estimator == "blackbox" else 0,
                         out_dev=ptr(self.out), noise_out_dev=ptr(noise_out), indices_out_dev=ptr(indices_out),
                         fvalue_out_dev=ptr(fvalue_out), logq_out_dev=ptr(logq_out),
                         workspace_dev=ptr(self.workspace(n_local)), stream=self._stream(),
                         f_weight_dev=ptr(f_weight), q_weight_dev=ptr(q_weight))

    def _identity_cfg(self):
        return native.make_opt_cfg("SGD", lr=0.0)

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
        rows = n_local * p.batch_size
        noise_t = self._noise_tensor(noise, base, n_local)
        idx_t = self._indices_tensor(minibatch, base, n_local)
        noise_o = torch.empty((rows, p.latent_dim), device=dev) if want_noise else None
        idx_o = torch.empty((n_local, p.batch_size), device=dev, dtype=torch.int32) if want_indices else None
        fvals = torch.empty(rows, device=dev) if want_fvalues else None
        logq = torch.empty(rows, device=dev) if want_fvalues else None
        args = self._args(n_local, number_samples, base, noise_t, idx_t, seed, offset, noise_o, idx_o, fvals, logq)
        native.check(self.lib.bsvi_amort_fwd_bwd(self.handle, C.byref(args)))
        engine.allreduce_sums(self.out)
        engine.check_exchange(self.device, self.params)
        # sums -> loss and gradients: the ELBO estimate is a mean over N*B rows (gradient_estimators.py:36,44)
        self._finalize(number_samples * p.batch_size)
        res = dict(loss=self.out[2], finite=self.out[3], nonfinite_count=self.out[1],
                   grads=self.out[OUT_HEADER:OUT_HEADER + p.n_params], n_local=n_local, sample_base=base)
        if want_noise:
            res["noise"] = noise_o
        if want_indices:
            res["indices"] = idx_o
        if want_fvalues:
            res["f"] = fvals
            res["logq"] = res["lq"] = logq
        return res

    def _finalize(self, divisor):
        """sums -> loss and gradients (an optimizer step with an empty mask: nothing moves)"""
        dev, p = self.device, self.program
        cfg = self._identity_cfg()
        zero_mask = getattr(self, "_zero_mask", None)
        if zero_mask is None:
            zero_mask = self._zero_mask = torch.zeros(max(p.n_params, 1), dtype=torch.uint8, device=dev)
            self._state0 = torch.zeros(4 * max(p.n_params, 1), device=dev)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        native.check(self.lib.bsvi_finalize_step(C.byref(cfg), ptr(self.params), ptr(self.out), ptr(self._state0),
                                                 ptr(zero_mask), p.n_params, int(divisor), None, None, self._stream()))
        self.grads_valid = True

    def evaluate_weighted(self, number_samples, f_weight, q_weight, seed, offset, noise=None, minibatch=None):
        """The second pass of a user-defined gradient estimator (`engine.custom_estimator_loss`), as
        `CompiledELBO.evaluate_weighted` with one weight per ROW ([number_samples, batch_size]): the draw and the minibatches of
        (seed, offset) again, -(sum_r a_r grad f_r + b_r grad log q_r) in the output block (bsvi_amort_args::f_weight_dev /
        q_weight_dev)."""
        from brancher_amd import engine
        rank, world = engine.dist_info()
        base, n_local = engine.shard(number_samples, rank, world)
        B = self.program.batch_size
        a = f_weight.reshape(-1)[base * B:(base + n_local) * B].contiguous().float()
        b = q_weight.reshape(-1)[base * B:(base + n_local) * B].contiguous().float()
        noise_t = self._noise_tensor(noise, base, n_local)
        idx_t = self._indices_tensor(minibatch, base, n_local)
        args = self._args(n_local, number_samples, base, noise_t, idx_t, seed, int(offset), f_weight=a, q_weight=b)
        native.check(self.lib.bsvi_amort_fwd_bwd(self.handle, C.byref(args)))
        engine.allreduce_sums(self.out)
        engine.check_exchange(self.device, self.params)
        self._finalize(1)
        return self.out[OUT_HEADER:OUT_HEADER + self.program.n_params]

    def _apply(self, network, rows, key, outputs):
        p = self.program
        if key not in outputs:
            raise KeyError("the network has no output %r (it has %s)" % (key, sorted(map(str, outputs))))
        x = torch.as_tensor(np.asarray(rows, dtype=np.float32)) if not torch.is_tensor(rows) else rows.float()
        x = x.reshape(x.shape[0], -1).contiguous().to(self.device)
        layers = p.enc_layers if network == 0 else p.dec_layers
        value, col = outputs[key]
        producer = next(l for l in layers if l.out_value == value)
        width = producer.n_out
        if producer.parts is not None:                      # merged heads: this key's own columns
            width = producer.parts[0].n_out if col == 0 else producer.parts[1].n_out
        full = torch.empty((x.shape[0], producer.n_out), device=self.device)
        ws = self.workspace((x.shape[0] + p.batch_size - 1) // p.batch_size)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        native.check(self.lib.bsvi_amort_apply(self.handle, network, ptr(self.params), ptr(x), x.shape[0], value,
                                               ptr(full), ptr(ws), self._stream()))
        return full[:, col:col + width].contiguous()

    def decode(self, z, key=None):
        """decoder(z)[key] on the current parameters: the posterior-predictive step of `examples/VAE_playground.py:90-103`
        (`model.get_sample(1, input_values={z: ...})["decoder_output"]`), rows [n, latent_dim] -> [n, width]"""
        return self._apply(1, z, self.program.logits_key if key is None else key, self.program.dec_outputs)

    def encode(self, x, key="mean"):
        """encoder(x)[key] on rows [n, n_features] (the amortised posterior's parameters for new data)"""
        return self._apply(0, x, key, self.program.enc_outputs)

    def named_grads(self):
        g = self.out[OUT_HEADER:].detach().cpu().numpy()
        return {par.name: g[off:off + size].reshape(par.shape).copy() for par, off, size, _ in self.program.parameters}

    def named_params(self):
        t = self.params.detach().cpu().numpy()
        return {par.name: t[off:off + size].reshape(par.shape).copy() for par, off, size, _ in self.program.parameters}

    def train(self, number_iterations, number_samples, optimizer="Adam", noise_seq=None, minibatch_seq=None, seed=None,
              pretraining_iterations=0, allow_persistent=True, _force_sharded_path=False, **opt_params):
        """the loop of `inference.py:95-108`: one bsvi_amort_fwd_bwd, (all-reduce), one fused finalize + finite
        check + optimizer step + loss log per iteration; nothing returns to the host inside the loop"""
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
        # Several ranks: the decoder's gradients are complete before the encoder's backward pass starts (bsvi_amort_bucket).  Their
        # range of the output block is reduced by the library on a stream of its own and all-reduced THERE, beside the encoder's
        # backward pass; the header and the other gradients follow at the end (BSVI_AMORT_BUCKETS=0: one all-reduce of the block).
        pieces = self._bucket_pieces() if (world > 1 or _force_sharded_path) else None
        if pieces is not None:
            native.check(self.lib.bsvi_amort_set_bucket_stream(self.handle, C.c_void_p(self._bucket_stream.cuda_stream)))
        try:
            for it in range(K):
                nz = None if noise_seq is None else self._noise_tensor(noise_seq[it], base, n_local)
                mb = None if minibatch_seq is None else self._indices_tensor(minibatch_seq[it], base, n_local)
                args = self._args(n_local, number_samples, base, nz, mb, seed, offset0 + it)
                mask = self.mask_all if it > pretraining_iterations else self.mask_first
                if pieces is not None:
                    self._bucket_stream.wait_stream(torch.cuda.current_stream(dev))      # (the last step's finalize read the block)
                native.check(self.lib.bsvi_amort_fwd_bwd(self.handle, C.byref(args)))
                if pieces is not None:
                    early, late = pieces
                    with torch.cuda.stream(self._bucket_stream):
                        engine.allreduce_sums(early)
                    for piece in late:
                        engine.allreduce_sums(piece)
                    torch.cuda.current_stream(dev).wait_stream(self._bucket_stream)
                elif world > 1 or _force_sharded_path:
                    engine.allreduce_sums(self.out)
                self._finalize_step(cfg, state, mask, number_samples * p.batch_size, loss_curve, finite, it)
        finally:
            if pieces is not None:
                native.check(self.lib.bsvi_amort_set_bucket_stream(self.handle, None))
        self.last_mode = ("stepwise" if world == 1 else "stepwise+allreduce") + ("+bucket" if pieces is not None else "")
        if world > 1:
            engine.check_exchange(self.device, self.params)         # (an abandoned exchange poisoned a step: say so, loudly)
        return loss_curve[:K], finite[:K]

    def _bucket_pieces(self):
        """(the decoder's range of the output block, [the rest as one or two views]) when the decoder's parameters are one range of
        the parameter vector and the collective is torch.distributed's / RCCL's (the one-shot exchange takes whole blocks); else None"""
        from brancher_amd import engine
        if os.environ.get("BSVI_AMORT_BUCKETS", "1") == "0" or engine.collective_kind() not in ("torch", "rccl"):
            return None
        first, count = C.c_uint32(), C.c_uint32()
        native.check(self.lib.bsvi_amort_bucket(self.handle, C.byref(first), C.byref(count)))
        if not count.value:
            return None
        if getattr(self, "_bucket_stream", None) is None:
            self._bucket_stream = torch.cuda.Stream(device=self.device)
        lo, hi = native.OUT_HEADER + first.value, native.OUT_HEADER + first.value + count.value
        late = [v for v in (self.out[:lo], self.out[hi:]) if v.numel()]
        return self.out[lo:hi], late

    def _finalize_step(self, cfg, state, mask, divisor, loss_curve, finite, it):
        p = self.program
        ptr = lambda t: C.c_void_p(t.data_ptr())
        native.check(self.lib.bsvi_finalize_step(
            C.byref(cfg), ptr(self.params), ptr(self.out), ptr(state), ptr(mask), p.n_params,
            divisor, C.c_void_p(loss_curve.data_ptr() + 4 * it),
            C.c_void_p(finite.data_ptr() + 4 * it), self._stream()))


# every native call of a compiled program runs with its device current (engine._bound_to_device)
from brancher_amd import engine as _engine  # noqa: E402  (engine imports this module lazily)
CompiledAmortized = _engine._bound_to_device(CompiledAmortized)
