"""
Engine: owns the device buffers of a compiled (joint, posterior) pair and drives the C ABI.

PyTorch-ROCm is used here for exactly three things: device memory (``torch.empty(...,
device='cuda')``), the current HIP stream, and ``torch.distributed`` (RCCL) for the one
all-reduce per step of the sample-sharded multi-GPU path.  All arithmetic of the hot path
happens inside libbsvi.so.

Multi-GPU: Monte-Carlo samples are i.i.d. given the parameters, so rank g
evaluates samples [base_g, base_g + n_g) — Philox counters are global sample indices, so the
union of the shards is bit-identical to a single-GPU run of the same seed — and the
per-rank sums (loss sum, non-finite count, gradient sums; 4 + P floats) are combined by ONE
``all_reduce(SUM)`` before the (replicated, identical) optimizer step.  No other collective
exists on the path.
"""
import ctypes as C
import os
import warnings

import numpy as np
import torch

from brancher_amd import config
from brancher_amd import lowering
from brancher_amd import native
from brancher_amd.native import ElboArgs, OUT_HEADER


class _CaptureRefused(Exception):
    """the HIP-graph capture of a sharded step was refused on some rank: every rank steps eagerly (`CompiledELBO._train_graph`)"""


class _Stale(Exception):
    """a prepared training call no longer matches the object's buffers"""


def _device():
    dev = config.get_device()
    if dev.type != "cuda":
        raise native.NativeError("brancher_amd evaluates models on an MI355X only (config.device is {!r}); "
                                 "there is no CPU execution path".format(str(dev)))
    return dev


def dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard(n_global, rank, world):
    """[base, base + n_local) of the global sample range owned by `rank`."""
    q, r = divmod(n_global, world)
    n_local = q + (1 if rank < r else 0)
    base = rank * q + min(rank, r)
    return base, n_local


_exchanges = {}          # device index -> collective.Exchange, or False: decided (by all ranks together) that RCCL serves
_exchange_generation = [0]      # bumped whenever an Exchange is created, replaced or closed: a kept HIP graph holds the peers' IPC
                                # region pointers of the exchange it was captured with BY VALUE (`_train_graph` keys on this)


def _exchange_changed():
    _exchange_generation[0] += 1


def collective_kind():
    """BSVI_COLLECTIVE, default `torch`: torch.distributed's all-reduce (RCCL over xGMI under the "nccl" backend).  The library's
    own one-shot exchange (`auto` / `exchange`) is OPT-IN: it has never run between two physical GPUs (every two-rank test
    shares one GPU), its device-side wait is bounded, and an abandoned call is an error (`check_exchange`) where RCCL would
    simply have waited for a rank that was writing a checkpoint or compiling a program."""
    return os.environ.get("BSVI_COLLECTIVE", "torch")


def _exchange_for(out):
    """The one-shot exchange of this device for a message of out.numel() floats, or None.  Decided ONCE per device and message
    capacity, by all ranks together and outside any stream capture: every rank creates its region and maps its peers'
    (HIP IPC), then a self-test all-reduce of known vectors must come back exact on every rank — the ranks vote with one
    torch.distributed all-reduce.  Anything short of that (no IPC access to a peer, no fine-grained memory, a wrong or late
    sum) and every rank uses RCCL for the rest of the process."""
    import torch.distributed as dist
    from brancher_amd import collective
    key = out.device.index
    ex = _exchanges.get(key)
    if ex is False:
        return None
    if ex is not None and ex.capacity >= out.numel():
        return ex
    if torch.cuda.is_current_stream_capturing():
        return None                 # (cannot be decided inside a capture: the graph path makes an untimed call first)
    if ex is not None:
        ex.close()
    _exchange_changed()
    capacity = max(out.numel(), 1024)
    ok = 1.0
    try:
        ex = collective.Exchange(capacity, device=out.device)
        ex.capacity = capacity
        if not ex.self_test():
            ok = 0.0
    except (native.NativeError, RuntimeError):
        ex, ok = None, 0.0
    vote = torch.tensor([ok], device=out.device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(vote, op=dist.ReduceOp.MIN)
    if float(vote.item()) < 1.0:
        if ex is not None:
            ex.close()
        _exchanges[key] = False
        return None
    _exchanges[key] = ex
    return ex


def loop_exchange(out):
    """The exchange for the in-kernel training loop of a sharded run (`bsvi_train_persistent_exchange`), or None: the one
    `allreduce_sums` uses between launches — decided once, by all ranks together — or, for ONE rank walking the sharded path
    (`_force_sharded_path` with BSVI_LOOP_EXCHANGE=force: tests), an exchange of its own.  BSVI_LOOP_EXCHANGE=0 keeps the
    launch-per-step sequence."""
    if os.environ.get("BSVI_LOOP_EXCHANGE", "1") == "0" or collective_kind() not in ("auto", "exchange"):
        return None
    if not out.is_cuda or out.numel() > 16384:
        return None
    if dist_info()[1] > 1:
        return _exchange_for(out)
    if os.environ.get("BSVI_LOOP_EXCHANGE") != "force":       # (one rank: the HIP-graph replay of the step sequence is the default)
        return None
    key = out.device.index
    ex = _exchanges.get(key)
    if ex is None or (ex and ex.capacity < out.numel()):
        from brancher_amd import collective
        try:
            ex = collective.Exchange(max(out.numel(), 1024), device=out.device)
            ex.capacity = max(out.numel(), 1024)
        except (native.NativeError, RuntimeError):
            ex = False
        _exchanges[key] = ex
        _exchange_changed()
    return ex or None


def module_links_of(*models):
    """every `torch.nn.Module` link (functions.ModuleLink) the models' link expressions call, once each"""
    from brancher_amd import symbolic as sym
    from brancher_amd.functions import ModuleLink
    found = []

    def walk(e):
        if not isinstance(e, sym.Expr):
            return
        if e.op == "call" and isinstance(e.attr[0], ModuleLink) and e.attr[0] not in found:
            found.append(e.attr[0])
        for a in e.args:
            walk(a)

    for model in models:
        if model is None:
            continue
        for v in model.flatten():
            link = getattr(v, "link", None)
            if link is not None and hasattr(link, "expressions"):
                for l in link.expressions().values():
                    walk(getattr(l, "expr", None))
    return found


def allreduce_sums(out):
    """The ONE collective of the multi-GPU path: sum the per-rank output blocks
    [loss sum, non-finite count, -, -, gradient sums...] over the sample shards.

    BSVI_COLLECTIVE=torch (the default): torch.distributed's all-reduce (RCCL over xGMI when the backend is "nccl"; gloo in the
    CPU tests).  BSVI_COLLECTIVE=rccl is the same RCCL call through the C ABI (`bsvi_allreduce` on torch's communicator).
    BSVI_COLLECTIVE=auto / exchange (opt-in, see `collective_kind`): device messages of up to 16384 floats — every message of
    the scalar and dense paths, 188 bytes at BASELINE config 1 — go through the library's one-shot direct-write all-reduce over
    IPC-mapped peer regions (`bsvi_exchange_*`: one one-workgroup kernel, no ring) when `_exchange_for` found it usable,
    everything else through torch.distributed."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return out
    kind = collective_kind()
    if kind in ("auto", "exchange") and out.is_cuda and out.numel() <= 16384 and out.dtype == torch.float32:
        ex = _exchange_for(out)
        if ex is not None:
            ex.allreduce(out)
            return out
    if kind == "rccl" and out.is_cuda and out.dtype == torch.float32:
        from brancher_amd import collective
        comm = collective.rccl_comm_ptr()
        if comm is not None:
            return collective.rccl_allreduce(out, comm)
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out


def check_exchange(device, params=None):
    """At the end of every public evaluation / training call of the three engines when ranks exchanged through the one-shot
    exchange: did every call meet its peers?  An abandoned call leaves NaN in the loss sum so that the optimizer step is
    skipped — but not provably on EVERY rank: a peer that had already collected all flags when the abort word was raised
    completes that call with valid totals and steps (csrc/collective.hip: one relaxed read of the abort word), so the
    replicas can be one optimizer step apart afterwards.  Hence: the ranks agree on the outcome with one torch.distributed
    all-reduce (every rank reaches this point: it is the end of a public call), and when any of them gave up, every rank
    takes rank 0's parameters again before every rank raises."""
    ex = _exchanges.get(device.index if hasattr(device, "index") else device)
    if not ex:
        return
    gave_up = ex.status()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        flag = torch.tensor([float(gave_up)], device=ex.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        gave_up = int(flag.item())
        if gave_up and params is not None:
            broadcast_from_rank0(params)
    if gave_up:
        raise native.NativeError("the one-shot exchange was abandoned at its call {} on some rank (a peer did not arrive within "
                                 "BSVI_EXCHANGE_TIMEOUT_MS, or raised the abort word); that call's sums are NaN where it was "
                                 "abandoned, the replicas may be one optimizer step apart — the parameters were taken from rank 0 "
                                 "again".format(gave_up))


def estimator_name(gradient_estimator):
    if gradient_estimator is None:
        return "pathwise"
    if isinstance(gradient_estimator, str):
        return gradient_estimator
    name = getattr(gradient_estimator, "kernel_name", None)
    if name is None:
        raise NotImplementedError("gradient estimator {!r} is not implemented by the fused kernel"
                                  .format(gradient_estimator))
    return name


def noise_from_named(program, named, n):
    """{variable name: array in the reference layout [N, B, d...]}  ->  [n_noise, N] fp32
    (structure-of-arrays, sample axis fastest)."""
    out = np.zeros((program.n_noise, n), dtype=np.float32)
    for name, slot in program.slot_by_name.items():
        if name not in named:
            raise KeyError("no noise supplied for latent variable {!r}".format(name))
        a = np.asarray(named[name], dtype=np.float32)
        a = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(-1, 1)
        if a.shape[0] == 1 and n > 1:
            a = np.repeat(a, n, axis=0)
        if a.shape[1] == 1 and slot.size > 1:
            a = np.repeat(a, slot.size, axis=1)
        if a.shape != (n, slot.size):
            raise ValueError("noise for {!r} has shape {}, expected ({}, {})".format(name, a.shape, n, slot.size))
        out[slot.base:slot.base + slot.size, :] = a.T
    return out


class FusedLoss:
    """What ``estimate_log_model_evidence(for_gradient=True)`` / ``InferenceMethod.compute_loss`` return: the value of one
    fused ELBO evaluation.  The kernel launch that produced the value also produced the gradients (of the LOSS, -ELBO,
    scaled by 1/N, in the compiled program's output block), so ``backward()`` only records how the value the user holds
    relates to them: ``grad_scale`` = d(this value) / d(loss).  Negation and multiplication by a number keep the handle —
    a hand-written loop in the style of `brancher/inference.py:95-108`
    (``loss = -model.estimate_log_model_evidence(...); loss.backward(); optimizer.update()``) drives the device
    optimizer with the right sign."""

    def __init__(self, compiled, tensor, grad_scale=1.0):
        self.compiled = compiled
        self.tensor = tensor          # 0-d device tensor
        self.grad_scale = float(grad_scale)

    def backward(self):
        if self.compiled is not None:
            self.compiled.pending_grad_scale = self.grad_scale
        return None

    def _scaled(self, c):
        return FusedLoss(self.compiled, self.tensor * c, self.grad_scale * c)

    def __neg__(self):
        return self._scaled(-1.0)

    def __mul__(self, c):
        return self._scaled(float(c))

    __rmul__ = __mul__

    def __truediv__(self, c):
        return self._scaled(1.0 / float(c))

    def detach(self):
        return self.tensor.detach()

    def item(self):
        return float(self.tensor.item())

    def __float__(self):
        return self.item()

    def cpu(self):
        return self.tensor.cpu()


def _bound_to_device(cls):
    """Every native call of a compiled program runs with the program's device current: the library allocates its tables
    and launches on the CURRENT HIP device, while buffers and stream belong to `self.device` — `config.set_device('cuda:1')`
    without a `torch.cuda.set_device(1)` must not split them."""
    import functools

    def guard(fn):
        @functools.wraps(fn)
        def on_device(self, *args, **kwargs):
            dev = getattr(self, "device", None) or kwargs.get("device") or _device()
            # (entering torch.cuda.device() costs ~10 us of host time even when nothing changes: a training call of
            #  20 iterations is ~100 us in all)
            if dev.index is not None and torch.cuda.current_device() == dev.index:
                return fn(self, *args, **kwargs)
            with torch.cuda.device(dev):
                return fn(self, *args, **kwargs)
        return on_device

    for name in ("__init__", "evaluate", "train", "_train_graph", "workspace", "optimizer_step", "decode", "encode", "_apply"):
        if name in cls.__dict__:
            setattr(cls, name, guard(cls.__dict__[name]))
    return cls


_group_seen = [None, 0]         # weak reference to the default process group last seen, and how many different ones there have been


def _process_group_identity():
    """what a captured all-reduce is tied to: (rank, world size, which default process group this is) — a graph kept from an
    earlier group must not be replayed in a later one.  The group is told apart by a GENERATION counted here against a weak
    reference (after destroy_process_group() / init_process_group() a new group object can get the old one's id())."""
    import weakref
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return (0, 1, 0)
    group = dist.group.WORLD
    seen = _group_seen[0]() if _group_seen[0] is not None else None
    if seen is not group:
        try:
            _group_seen[0] = weakref.ref(group)
        except TypeError:                       # (not weak-referenceable: hold it — the comparison above stays exact)
            _group_seen[0] = (lambda g: (lambda: g))(group)
        _group_seen[1] += 1
    return (dist.get_rank(), dist.get_world_size(), _group_seen[1])


def _all_ranks_agree(ok, device):
    """True when `ok` holds on EVERY rank: one MIN all-reduce of a flag (no process group: this rank's own answer)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return bool(ok)
    flag = torch.tensor([1.0 if ok else 0.0], device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return float(flag.item()) >= 1.0


def broadcast_from_rank0(tensor):
    """ranks must start from identical parameters (and draw with the same seed): every rank keeps its own copy and only
    sums are all-reduced, so nothing would ever re-synchronise them"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(tensor, 0)
    return tensor


def shared_seed(seed, device):
    """the Philox key of a call: the caller's seed, else torch's initial seed — rank 0's on every rank"""
    value = int(torch.initial_seed() if seed is None else seed) & 0x7FFFFFFFFFFFFFFF
    import torch.distributed as dist
    if seed is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([value], dtype=torch.int64, device=device)
        dist.broadcast(t, 0)
        value = int(t.item())
    return value


def optimizer_step(store, cfg, state, mask):
    """one device optimizer step of the parameters selected by `mask` from the gradients in the store's output block
    (`bsvi_optimizer_step`): what ``ProbabilisticOptimizer.update()`` runs"""
    scale = getattr(store, "pending_grad_scale", 1.0)
    if scale != 1.0:
        store.out[OUT_HEADER:] *= scale          # the value the user called backward() on was scale * loss
        store.pending_grad_scale = 1.0
    ptr = lambda t: C.c_void_p(t.data_ptr())
    with torch.cuda.device(store.device):
        native.check(store.lib.bsvi_optimizer_step(C.byref(cfg), ptr(store.params), ptr(store.out), ptr(state), ptr(mask),
                                                   store.n_params, store._stream()))


@_bound_to_device
class CompiledELBO:
    def __init__(self, joint_model, posterior_model, estimator="pathwise", device=None, program=None,
                 bind_parameters=True):
        self.device = device or _device()
        self.program = program if program is not None else lowering.lower(joint_model, posterior_model, estimator)
        self.native = native.NativeProgram(self.program)
        self.lib = self.native.lib
        p = self.program
        dev = self.device
        self.n_params = p.n_params
        self.params = broadcast_from_rank0(torch.from_numpy(p.initial_params()).to(dev))
        self.obs = torch.from_numpy(np.ascontiguousarray(p.obs)).to(dev) if p.obs.size else torch.zeros(1, device=dev)
        self.out = torch.zeros(OUT_HEADER + max(p.n_params, 1), device=dev, dtype=torch.float32)
        self.grads_valid = False
        active = np.ascontiguousarray(p.param_active, dtype=np.uint8)
        group = p.param_group
        first_group = 0 if np.any(active[group == 0]) else 1
        self.mask_all = torch.from_numpy(active.copy()).to(dev)
        self.mask_first = torch.from_numpy((active * (group == first_group)).astype(np.uint8)).to(dev)
        # batched multivariate-normal terms (lowering.mvn_external): the BASE program (this one without those terms; it makes
        # the draw), one bsvi_mvn node per term, and the program's noise tensor whose last rows the nodes fill
        self._externals = []
        if getattr(p, "externals", None):
            self.base_program = lowering.lower(joint_model, posterior_model, estimator, external="omit")
            if [(q.name, o, n) for q, o, n, _ in self.base_program.parameters] != [(q.name, o, n) for q, o, n, _ in p.parameters]:
                raise RuntimeError("parameter layouts of the base program and the full program differ")
            self.base_native = native.NativeProgram(self.base_program)
            bp = self.base_program
            self.base_obs = torch.from_numpy(np.ascontiguousarray(bp.obs)).to(dev) if bp.obs.size else torch.zeros(1, device=dev)
            self.base_out = torch.zeros(OUT_HEADER + max(bp.n_params, 1), device=dev, dtype=torch.float32)
            # (round 6: a node is a batched multivariate-normal term or a REDUCE node — lowering.reduce_external)
            self._externals = [native.ReduceNode(e) if getattr(e, "kind", "mvn") == "reduce" else native.MvnNode(e) for e in p.externals]
            self._base_buffers = {}
        # observations that are a MINIBATCH of a dataset (scalar-path f-1, lowering.MinibatchObs): the datasets resident on the device,
        # the stretches of the observation buffer refreshed in front of every evaluation (`_refresh_minibatches`)
        self._minibatches = []
        for mb in getattr(p, "minibatches", None) or []:
            if self._externals:
                raise lowering.LoweringError("minibatch observations beside batched multivariate-normal terms are not lowered yet")
            data = torch.from_numpy(np.ascontiguousarray(mb["dataset"], dtype=np.float32).reshape(mb["dataset_size"], -1)).to(dev)
            self._minibatches.append(dict(mb, data=data))
        self._workspaces = {}
        self._train_plans = {}
        self._fast_train = {}
        self._plain_args = {}
        self._shares_for = None
        self._noise_cache = None
        self.iteration = 0          # Philox counter offset: never reuse noise across calls
        if bind_parameters:
            for par, off, size, _ in p.parameters:
                par.bind(self, off)

    # ---- ParameterStore protocol (modules.Parameter) -------------------------------------
    def read_params(self, offset, size):
        return self.params[offset:offset + size].detach().cpu().numpy()

    def write_params(self, offset, values):
        self.params[offset:offset + values.size] = torch.from_numpy(np.ascontiguousarray(values)).to(self.device)

    def read_grads(self, offset, size):
        if not self.grads_valid:
            return None
        o = OUT_HEADER + offset
        return self.out[o:o + size].detach().cpu().numpy()

    # ---- helpers ------------------------------------------------------------------------------
    def workspace(self, n_local):
        ws = self._workspaces.get(n_local)
        if ws is None:
            nbytes = self.native.workspace_bytes(n_local)
            if nbytes == 0:
                raise native.NativeError("the model program does not fit the LDS budget for {} samples".format(n_local))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._workspaces[n_local] = ws
        return ws

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _seed(self, seed):
        """the Philox key of a call (rank 0's on every rank): a collective + a host sync when `seed` is None on several
        ranks, so `evaluate` / `train` resolve it once, up front, and hand the integer down"""
        return shared_seed(seed, self.device)

    def _resolved(self, seed):
        if seed is None:
            _, world = dist_info()
            if world > 1:
                raise native.NativeError("internal: the seed of a multi-rank call must be resolved before the launch path "
                                         "(CompiledELBO._seed), never inside it")
            return shared_seed(None, self.device)
        return int(seed) & 0x7FFFFFFFFFFFFFFF

    def _elbo_args(self, n_local, n_global, base, noise=None, seed=None, offset=0, samples_out=None,
                   noise_out=None, fvalue_out=None):
        if noise is None and samples_out is None and noise_out is None and fvalue_out is None:
            # the plain call (Philox noise, no per-sample outputs): one struct per shard, only what changes is set again
            key = (n_local, n_global, base)
            args = self._plain_args.get(key)
            if args is None:
                args = self._plain_args[key] = self._elbo_args(n_local, n_global, base, None, seed, offset, None, None,
                                                               torch.empty(0))     # (any non-None output: builds the struct below)
                args.fvalue_out_dev = None
            # (seed is an integer here: callers resolve `None` ONCE per call, outside any stream capture — resolving it
            #  is a broadcast + a host sync on several ranks)
            args.seed, args.offset, args.stream = self._resolved(seed), int(offset), self._stream()
            args.offset_dev = None
            return args
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        return ElboArgs(params_dev=ptr(self.params), obs_dev=ptr(self.obs), noise_dev=ptr(noise),
                        seed=self._resolved(seed), offset=int(offset), n_samples_local=n_local,
                        n_samples_global=n_global, sample_base=base, out_dev=ptr(self.out),
                        samples_out_dev=ptr(samples_out), noise_out_dev=ptr(noise_out),
                        fvalue_out_dev=ptr(fvalue_out), workspace_dev=ptr(self.workspace(n_local)),
                        stream=self._stream())

    def _external_rows(self, n_local, n_global, base, noise_t, seed, offset, minibatch=None):
        """Models with batched multivariate-normal terms, per evaluation: (1) the base program draws the posterior's sample
        (its noise and slot values land in buffers), (2) every bsvi_mvn node reads the draw and writes the rows of its linear
        surrogate — log p and d log p / d inputs of each sample — behind the real rows of the noise tensor, (3) the caller
        launches the full program on that tensor.  Returns the tensor [n_noise, n_local]."""
        p, bp, dev = self.program, self.base_program, self.device
        bufs = self._base_buffers.get(n_local)
        if bufs is None:
            nbytes = self.base_native.workspace_bytes(n_local)
            if nbytes == 0:
                raise native.NativeError("the base program does not fit the LDS budget for {} samples".format(n_local))
            bufs = self._base_buffers[n_local] = dict(ws=torch.empty(nbytes, dtype=torch.uint8, device=dev),
                                                      samples=torch.empty((bp.n_noise, n_local), device=dev),
                                                      noise=torch.zeros((p.n_noise, n_local), device=dev))
        # (supplied noise: a COPY carries the surrogate rows — the caller's tensor may be replayed on another model or estimator)
        full = bufs["noise"]
        if noise_t is not None:
            full.copy_(noise_t)
            noise_t = full
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        args = ElboArgs(params_dev=ptr(self.params), obs_dev=ptr(self.base_obs), noise_dev=ptr(noise_t),
                        seed=self._resolved(seed), offset=int(offset), n_samples_local=n_local, n_samples_global=n_global,
                        sample_base=base, out_dev=ptr(self.base_out), samples_out_dev=ptr(bufs["samples"]),
                        noise_out_dev=None if noise_t is not None else ptr(full), fvalue_out_dev=None,
                        workspace_dev=ptr(bufs["ws"]), stream=self._stream())
        self.base_native.attach_shares()
        native.check(self.lib.bsvi_elbo_fwd_bwd(self.base_native.handle, C.byref(args)))
        for node in self._externals:
            extra = {}
            if getattr(node.node, "kind", "mvn") == "reduce":
                # the reduce node's data — a variable observed by flag only — is drawn on the device from (seed, offset), the same on
                # every rank, or handed in with the minibatch rows (`minibatch`: {variable name: value}; parity tests replay the reference's)
                extra = dict(seed=self._resolved(seed), offset=int(offset))
                given = minibatch.get(node.node.drawn_name) if isinstance(minibatch, dict) and node.node.drawn_name else None
                if given is not None:
                    data = torch.from_numpy(np.ascontiguousarray(np.asarray(given, dtype=np.float32).reshape(-1))).to(dev)
                    if data.numel() != node.node.n_data * node.node.rows * node.node.cols:
                        raise ValueError("the value of {!r} must have {} x {} x {} elements".format(
                            node.node.drawn_name, node.node.n_data, node.node.rows, node.node.cols))
                    bufs["given_data"] = data            # (alive until the launches that read it have run: one evaluation at a time)
                    extra["data_ptr"] = ptr(data)
            node.eval(ptr(self.params), ptr(bufs["samples"]), C.c_void_p(full.data_ptr() + 4 * node.node.row0 * n_local),
                      n_local, self._stream(), **extra)
        return full

    def _refresh_minibatches(self, seed, offset, minibatch=None, want_indices=False):
        """The minibatch data path of the scalar engine (`standard_variables.py:71-112`, `distributions.py:393-473`): in front of an
        evaluation every stretch of the observation buffer that is a minibatch takes its rows for this (seed, offset) —
        `bsvi_minibatch_gather`: the keyed bijection of the dense path (sources that share a RandomIndices variable share the draw;
        an EmpiricalVariable with its own batch_size draws with its own key), or the caller's rows (`minibatch`: {indices name:
        rows}, parity tests replay the reference's).  Every rank draws the same rows: the key is (seed, offset)."""
        if not self._minibatches:
            return None
        used = {}
        for mb in self._minibatches:
            given = None
            if minibatch is not None:
                rows = minibatch.get(mb["indices_name"]) if isinstance(minibatch, dict) else minibatch
                if rows is not None:
                    idx = np.asarray(rows, dtype=np.int64).reshape(-1)
                    if idx.size != mb["batch"] or idx.min() < 0 or idx.max() >= mb["dataset_size"]:
                        raise ValueError("minibatch rows of {!r} must be {} rows in [0, {})".format(mb["indices_name"], mb["batch"], mb["dataset_size"]))
                    given = torch.from_numpy(idx.astype(np.int32)).to(self.device)
            out_idx = torch.empty(mb["batch"], dtype=torch.int32, device=self.device) if want_indices else None
            key = (int(seed) ^ (0x9E3779B97F4A7C15 * mb["group"])) & 0x7FFFFFFFFFFFFFFF       # (group 0: the dense path's key)
            native.check(self.lib.bsvi_minibatch_gather(
                C.c_void_p(mb["data"].data_ptr()), mb["dataset_size"], mb["row"], mb["batch"],
                C.c_void_p(given.data_ptr()) if given is not None else None, key, int(offset),
                C.c_void_p(self.obs.data_ptr() + 4 * mb["offset"]), C.c_void_p(out_idx.data_ptr()) if out_idx is not None else None,
                self._stream()))
            if want_indices:
                used[mb["indices_name"]] = out_idx
        return used

    def _noise_tensor(self, noise, n_global, base, n_local):
        """named dict / [n_noise, N] array / device tensor -> device [n_noise, n_local] (this rank's columns)"""
        if noise is None:
            return None
        if isinstance(noise, dict):
            noise = noise_from_named(self.program, noise, n_global)
        if isinstance(noise, np.ndarray):
            noise = torch.from_numpy(np.ascontiguousarray(noise[:, base:base + n_local], dtype=np.float32))
            return noise.to(self.device)
        if noise.shape[1] != n_local:
            noise = noise[:, base:base + n_local].contiguous()
        return noise

    # ---- one ELBO evaluation -------------------------------------------------------------------
    def evaluate(self, number_samples, noise=None, seed=None, offset=None, want_samples=False,
                 want_noise=False, want_fvalues=False, minibatch=None, want_indices=False):
        """Loss and gradients of one ELBO estimate (`variables.py:843-870` with
        for_gradient=True + `inference.py:100`).  Returns a dict of device tensors."""
        rank, world = dist_info()
        base, n_local = shard(number_samples, rank, world)
        if n_local == 0:
            raise ValueError("number_samples={} is smaller than the number of GPUs {}".format(number_samples, world))
        if offset is None:
            offset = self.iteration
            self.iteration += 1
        dev = self.device
        p = self.program
        seed = self._seed(seed)
        noise_t = self._noise_tensor(noise, number_samples, base, n_local)
        samples = torch.empty((p.n_noise, n_local), device=dev) if want_samples else None
        noise_o = torch.empty((p.n_noise, n_local), device=dev) if want_noise else None
        fvals = torch.empty((2, n_local), device=dev) if want_fvalues else None
        if self._externals:
            noise_t = self._external_rows(n_local, number_samples, base, noise_t, seed, offset, minibatch)
        used_rows = self._refresh_minibatches(seed, offset, minibatch, want_indices)
        args = self._elbo_args(n_local, number_samples, base, noise_t, seed, offset, samples, noise_o, fvals)
        self.native.attach_shares()
        native.check(self.lib.bsvi_elbo_fwd_bwd(self.native.handle, C.byref(args)))
        allreduce_sums(self.out)
        check_exchange(self.device, self.params)
        native.check(self.lib.bsvi_finalize(self.native.handle, C.c_void_p(self.out.data_ptr()), number_samples,
                                            self._stream()))
        self.grads_valid = True
        res = dict(loss=self.out[2], finite=self.out[3], nonfinite_count=self.out[1],
                   grads=self.out[OUT_HEADER:OUT_HEADER + p.n_params], n_local=n_local, sample_base=base)
        if want_indices and used_rows is not None:
            res["indices"] = used_rows
        if want_samples:
            res["samples"] = samples
        if want_noise:
            res["noise"] = noise_o
        if want_fvalues:
            res["f"], res["lq"] = fvals[0], fvals[1]
        return res

    def evaluate_weighted(self, number_samples, f_weight, q_weight, seed, offset, noise=None, minibatch=None):
        """The second pass of a user-defined gradient estimator (`custom_estimator_loss`): with the draw of (seed, offset)
        again, leave  -(sum_n a_n grad f_n + b_n grad log q_n)  in the output block — the gradient of loss = -g(f, log q) for
        a_n = dg/df_n, b_n = dg/dlog q_n (bsvi_elbo_args::f_weight_dev / q_weight_dev; the program must be a BlackBox one)."""
        rank, world = dist_info()
        base, n_local = shard(number_samples, rank, world)
        a = f_weight.reshape(-1)[base:base + n_local].contiguous().float()
        b = q_weight.reshape(-1)[base:base + n_local].contiguous().float()
        noise_t = self._noise_tensor(noise, number_samples, base, n_local)
        if self._externals:
            # batched multivariate-normal terms re-enter the program as linear surrogate records — model terms like any other,
            # so a_n weights them too; their rows come from the same three-launch sequence as in `evaluate`
            noise_t = self._external_rows(n_local, number_samples, base, noise_t, self._seed(seed), int(offset), minibatch)
        self._refresh_minibatches(self._seed(seed), int(offset), minibatch)
        args = ElboArgs.from_buffer_copy(self._elbo_args(n_local, number_samples, base, None, self._seed(seed), int(offset)))
        args.stream = self._stream()
        args.noise_dev = noise_t.data_ptr() if noise_t is not None else None
        args.f_weight_dev, args.q_weight_dev = a.data_ptr(), b.data_ptr()
        self.native.attach_shares()
        native.check(self.lib.bsvi_elbo_fwd_bwd(self.native.handle, C.byref(args)))
        allreduce_sums(self.out)
        check_exchange(self.device, self.params)
        native.check(self.lib.bsvi_finalize(self.native.handle, C.c_void_p(self.out.data_ptr()), 1, self._stream()))
        self.grads_valid = True
        return self.out[OUT_HEADER:OUT_HEADER + self.program.n_params]

    def named_grads(self):
        g = self.out[OUT_HEADER:].detach().cpu().numpy()
        out = {}
        for par, off, size, _ in self.program.parameters:
            out[par.name] = g[off:off + size].reshape(par.shape).copy()
        return out

    def named_params(self):
        t = self.params.detach().cpu().numpy()
        return {par.name: t[off:off + size].reshape(par.shape).copy() for par, off, size, _ in self.program.parameters}

    def samples_by_name(self, samples):
        """device [n_slots, n] -> {name: numpy [n, B, D1, D2]} (reference layout at the API edge)"""
        s = samples.detach().cpu().numpy()
        out = {}
        for name, slot in self.program.slot_by_name.items():
            out[name] = s[slot.base:slot.base + slot.size, :].T.reshape((s.shape[1],) + tuple(slot.shape)).copy()
        return out

    # ---- the optimisation loop --------------------------------------------------------------------
    def prefers_stepwise(self, number_samples):
        """With the model's records split into 4+ program shares the launch-per-iteration path (one wave per workgroup,
        `reduce_kernel` adding the rows of partial sums once) beats the persistent trainer, whose in-kernel exchange
        every workgroup walks in full: 29.0 vs 32.3 us per iteration at BASELINE config 1."""
        rank, world = dist_info()
        base, n_local = shard(number_samples, rank, world)
        if self.native.engine(n_local, 2)["engine"] == "specialised":
            return False        # the program-specialised kernel keeps the whole loop in one launch
        self.native.ensure_shares(n_local)
        self.native.attach_shares()
        self._shares_for = n_local
        return world == 1 and getattr(self.native, "_elbo_shares_set", 0) >= 4 \
            and os.environ.get("BSVI_ELBO_SHARES", "1") != "0"

    def _train_graph(self, K, n_local, n_global, base, cfg, state, loss_curve, finite, seed, offset0, pretraining):
        """The sharded iteration — `bsvi_elbo_fwd_bwd` (this rank's samples), ONE all-reduce of the [4+P] sums over the
        ranks (RCCL), `bsvi_finalize_step_counted` (loss, finite flag, replicated optimizer step) — captured in a HIP
        graph, `BSVI_GRAPH_UNROLL` iterations per graph, and replayed.  The iteration number (Philox offset, loss slot,
        pretraining mask) lives in device memory, so the replays are identical launches; the host only enqueues graphs.

        The graphs are KEPT for a repeat of the same call (round 5: every `train` captured anew — a warm-up launch, a
        synchronize, the capture and the instantiation, milliseconds in front of a 20-iteration call of ~0.6 ms): they own their
        loss / flag / optimizer-state buffers and counters; a repeat clears the state, sets the counters, replays and copies the
        curve out.  Keyed by everything the captured launches hold by value or by address."""
        dev, p = self.device, self.program
        ptr = lambda t: C.c_void_p(t.data_ptr())
        # (the exchange's generation: the captured exchange kernels hold the peers' IPC region pointers by value, and
        #  `_exchange_for` closes and replaces the exchange when a larger message arrives — ADVICE r5)
        key = (int(K), int(n_local), int(n_global), int(base), bytes(cfg), int(self._resolved(seed)), int(pretraining),
               self.params.data_ptr(), self.out.data_ptr(), collective_kind(), _exchange_generation[0], os.environ.get("BSVI_JIT"),
               _graph_unroll(), _process_group_identity())
        keep = os.environ.get("BSVI_GRAPH_KEEP", "1") != "0"       # 0: capture per call, nothing kept (a first multi-GPU run's way back)
        cache = self.__dict__.setdefault("_graph_cache", {})
        entry = cache.get(key) if keep else None
        if entry is False:
            # some rank could not capture this call before: EVERY rank recorded that (the vote below) and steps eagerly
            raise _CaptureRefused("capture of this call was refused on some rank before")
        if entry is None:
            # A miss issues one more collective than a hit (the warm-up below) and the vote — so hit or miss must be the same on
            # every rank: an entry is kept only when EVERY rank captured, otherwise every rank keeps a refusal (a rank
            # whose capture alone failed would otherwise be one all-reduce ahead of its peers on the next identical call).
            own_curve, own_finite, own_state = training_buffers(K, p.n_params, dev)
            counters = torch.tensor([int(offset0), 0], dtype=torch.int64, device=dev)
            failure = None
            # hiprtc / module loading and RCCL's first-call set-up cannot happen inside a capture: one untimed launch of each
            try:
                warm = self._elbo_args(n_local, n_global, base, None, seed, int(offset0))
                self.native.attach_shares()
                native.check(self.lib.bsvi_elbo_fwd_bwd(self.native.handle, C.byref(warm)))
            except (RuntimeError, native.NativeError) as err:
                failure = err
            allreduce_sums(torch.zeros_like(self.out))          # (every rank, whatever happened above: the collectives stay aligned)
            torch.cuda.synchronize(dev)
            own_cfg = type(cfg).from_buffer_copy(cfg)          # (the captured launches read the block at capture time only; kept anyway)

            def capture(n_steps):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    # a PRIVATE argument block (the cached one of the plain calls must not keep a pointer to `counters`),
                    # on the capturing stream
                    args = ElboArgs.from_buffer_copy(self._elbo_args(n_local, n_global, base, None, seed, 0))
                    args.stream = self._stream()
                    args.offset_dev = counters.data_ptr()
                    for _ in range(n_steps):
                        self.native.attach_shares()
                        native.check(self.lib.bsvi_elbo_fwd_bwd(self.native.handle, C.byref(args)))
                        allreduce_sums(self.out)
                        native.check(self.lib.bsvi_finalize_step_counted(
                            C.byref(own_cfg), ptr(self.params), ptr(self.out), ptr(own_state), ptr(self.mask_all), ptr(self.mask_first),
                            pretraining, p.n_params, n_global, ptr(own_curve), ptr(own_finite), ptr(counters), self._stream()))
                return graph

            unroll = min(K, _graph_unroll())
            if failure is None:
                try:
                    entry = dict(curve=own_curve, finite=own_finite, state=own_state, counters=counters, cfg=own_cfg, fresh=True,
                                 main=capture(unroll), n_main=K // unroll, tail=capture(K % unroll) if K % unroll else None)
                except (RuntimeError, native.NativeError) as err:
                    failure, entry = err, None
            if not _all_ranks_agree(failure is None, dev):
                entry = None
                if keep:
                    cache[key] = False
                raise _CaptureRefused(str(failure) if failure is not None else "a peer rank could not capture the step")
            if keep:
                while len(cache) >= 4:                          # (the executables, their pools and buffers live as long as the entry;
                    # an evicted one may still be replaying on the stream: it stays referenced until three more have gone)
                    evicted = cache.pop(next(iter(cache)))
                    if evicted:
                        self._graphs = getattr(self, "_graphs", [])[-3:] + [evicted]
                cache[key] = entry
            else:
                self._graphs = getattr(self, "_graphs", [])[-3:] + [entry]     # (alive until its replays have run)
        if not entry["fresh"]:
            # a repeat: fresh optimizer state, the call's first Philox offset, iteration counter 0 — stream-ordered fills
            entry["state"].zero_()
            entry["counters"].zero_()
            entry["counters"][0:1].fill_(int(offset0))
        entry["fresh"] = False
        for _ in range(entry["n_main"]):
            entry["main"].replay()
        if entry["tail"] is not None:
            entry["tail"].replay()
        loss_curve[:K].copy_(entry["curve"][:K])
        finite[:K].copy_(entry["finite"][:K])

    def _prepare_fast_train(self, n_local, n_global, base, cfg, pretraining):
        """a repeat of one in-kernel training call (single rank, Philox noise, fresh optimizer inside the kernel) as a
        closure over pre-built ctypes objects"""
        args = ElboArgs.from_buffer_copy(self._elbo_args(n_local, n_global, base, None, 0, 0))
        args.offset_dev = None
        fn = self.lib.bsvi_train_persistent2
        handle, p_args, p_cfg = self.native.handle, C.byref(args), C.byref(cfg)
        params, mask_all, mask_first = (C.c_void_p(t.data_ptr()) for t in (self.params, self.mask_all, self.mask_first))
        params_ptr = self.params.data_ptr()
        dev = self.device
        # (the handle of the current stream without building a torch.cuda.Stream object around it: 0.3 us instead of 2.2 of a
        #  call that is ~10 us in all — tools/r5/call_anatomy.py; the public call where this torch has no such entry)
        raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        current_stream = (lambda: raw_stream(dev_index)) if raw_stream is not None else (lambda: torch.cuda.current_stream(dev).cuda_stream)
        empty = torch.empty
        import torch.distributed as dist
        is_distributed = dist.is_initialized if dist.is_available() else (lambda: False)

        def run(K, seed):
            Ka = (K + 3) // 4 * 4
            # (a sibling estimator re-bound the parameter buffer, or a process group appeared since: the full path decides)
            if self.params.data_ptr() != params_ptr or is_distributed():
                self._fast_train.clear()
                return None
            if seed is None:
                seed = shared_seed(None, dev)      # (per call: a torch.manual_seed() since the last one counts, as on the full path)
            buf = empty(2 * Ka, device=dev)
            args.seed = int(seed) & 0x7FFFFFFFFFFFFFFF
            args.offset = self.iteration
            args.stream = current_stream()
            self.iteration += K
            rc = fn(handle, p_args, p_cfg, params, None, mask_all, mask_first, pretraining, K,
                    C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr() + 4 * Ka))
            if rc != 0:                                            # (e.g. the engine was switched by BSVI_JIT since)
                self.iteration -= K
                self._fast_train.clear()
                return None
            self.grads_valid = True
            self.last_mode = "persistent"
            if K == Ka:
                return buf.view(2, Ka).unbind(0)                   # (both views from one call)
            return buf[:K], buf[Ka:Ka + K]

        def call(K, seed):
            out = run(K, seed)
            if out is None:                                        # fall back to the full path once; it re-prepares
                raise _Stale()
            return out
        return call

    def train(self, number_iterations, number_samples, optimizer="Adam", noise_seq=None, seed=None,
              pretraining_iterations=0, allow_persistent=True, minibatch_seq=None, _force_sharded_path=False,
              **opt_params):
        """`brancher/inference.py:95-108` on the device.  Returns (loss_curve, finite_flags) as
        device tensors of length number_iterations; nothing synchronises with the host."""
        # The short call (the driver times 20 iterations: ~100 us on the device): everything a repeat of the same call needs
        # — optimizer block, argument block, ctypes pointers, the launch plan — is kept from the first one, and the call is
        # a buffer, five stores into the argument block and ONE library call.
        if noise_seq is None and minibatch_seq is None and not _force_sharded_path and allow_persistent:
            # (option values in a hashable normal form: betas=[0.9, 0.99] is as good as the tuple)
            fast_key = (int(number_samples), optimizer if isinstance(optimizer, str) else None, int(pretraining_iterations),
                        tuple(sorted((k, tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in opt_params.items())))
            try:
                fast = self._fast_train.get(fast_key)
            except TypeError:                   # an unhashable option value: no prepared repeat for this call
                fast_key, fast = None, None
            if fast is not None and number_iterations > 0:
                try:
                    return fast(int(number_iterations), seed)
                except _Stale:
                    pass
        else:
            fast_key = None
        cfg = native.make_opt_cfg(optimizer, **opt_params)
        rank, world = dist_info()
        base, n_local = shard(number_samples, rank, world)
        if n_local == 0:
            raise ValueError("number_samples={} is smaller than the number of GPUs {}".format(number_samples, world))
        broadcast_from_rank0(self.params)
        seed = self._seed(seed)
        dev = self.device
        p = self.program
        K = int(number_iterations)
        # (_force_sharded_path: run the multi-GPU step sequence on one GPU — tests)
        # which launch path serves this shard: decided once per (shard size, switches) — four library queries that a
        # short training call (the driver times 20 iterations) would otherwise repeat
        plan_key = (n_local, bool(allow_persistent), world, bool(_force_sharded_path), os.environ.get("BSVI_JIT"))
        plan = self._train_plans.get(plan_key)
        if plan is None or self._shares_for != n_local:
            self.native.ensure_shares(n_local)          # (the share set attached to the program follows the last shard size)
            self._shares_for = n_local
        if plan is None:
            # (observations that are a minibatch change in every iteration: the loop cannot stay in one launch)
            persistent = (allow_persistent and world == 1 and not _force_sharded_path and not self._externals
                          and not self._minibatches and self.native.persistent_supported(n_local))
            shares = self.native.split_shares(n_local) if persistent else None
            # the specialised in-kernel loop starts a fresh optimizer itself and nobody reads its final state: no state
            # buffer, and with it no fill launch in front of the training launch
            fresh = persistent and shares is None and self.native.engine(n_local, 2)["engine"] == "specialised"
            plan = self._train_plans[plan_key] = (persistent, shares, fresh)
        persistent, shares, fresh_in_kernel = plan
        loss_curve, finite, state = training_buffers(K, p.n_params, dev, with_state=not (fresh_in_kernel and K > 0))
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        noise_t = None
        if noise_seq is not None:
            # [K][n_noise][n_local] contiguous (the persistent trainer advances by n_noise*n_local per iteration)
            mats = [noise_from_named(p, nz, number_samples) if isinstance(nz, dict) else np.asarray(nz)
                    for nz in noise_seq]
            arr = np.ascontiguousarray(np.stack(mats)[:, :, base:base + n_local], dtype=np.float32)
            noise_t = torch.from_numpy(arr).to(dev)
        offset0 = self.iteration
        self.iteration += K
        self.grads_valid = True
        if K == 0:
            return loss_curve[:0], finite[:0]

        if persistent:
            args = self._elbo_args(n_local, number_samples, base, noise_t, seed, offset0)
            if shares is not None:
                # one wave per workgroup AND the model's log-prob records split over workgroups
                native.check(self.lib.bsvi_train_persistent_split(
                    self.native.handle, shares, len(shares), C.byref(args), C.byref(cfg), ptr(self.params), ptr(state),
                    ptr(self.mask_all), ptr(self.mask_first), int(pretraining_iterations), K, ptr(loss_curve), ptr(finite)))
            else:
                call = lambda st: self.lib.bsvi_train_persistent2(
                    self.native.handle, C.byref(args), C.byref(cfg), ptr(self.params), ptr(st), ptr(self.mask_all),
                    ptr(self.mask_first), int(pretraining_iterations), K, ptr(loss_curve), ptr(finite))
                rc = call(state)
                if rc == 0 and state is None and fast_key is not None and fast_key[1] is not None and K > 0 \
                        and os.environ.get("BSVI_FAST_TRAIN", "1") != "0":
                    self._fast_train[fast_key] = self._prepare_fast_train(n_local, number_samples, base, cfg,
                                                                          int(pretraining_iterations))
                if rc != 0 and state is None:
                    # the plan said "specialised in-kernel loop" before the lazy hiprtc compile; if that compile then
                    # fails the library leaves the program to the interpreter's trainer, which needs a state buffer:
                    # forget the plan and run this call (and every later one) with one
                    self._train_plans[plan_key] = (persistent, shares, False)
                    loss_curve, finite, state = training_buffers(K, p.n_params, dev, with_state=True)
                    rc = call(state)
                native.check(rc)
            self.last_mode = "persistent"
            return loss_curve, finite

        sharded = world > 1 or _force_sharded_path
        if sharded and noise_t is None and not self._externals and not self._minibatches:
            # several ranks, ONE launch each: the cross-rank sums are exchanged inside the in-kernel loop (spec_main.h,
            # spec_exchange).  Whether it serves — the shard on the specialised one-workgroup kernel, every parameter
            # owned by a thread of one wave, the exchange usable — is probed once per plan with an empty call (which
            # also compiles the kernel variant) and VOTED: a rank on another path would leave its peers waiting.
            xkey = plan_key + ("loop exchange",)
            xplan = self._train_plans.get(xkey)
            ex = loop_exchange(self.out)
            args = self._elbo_args(n_local, number_samples, base, None, seed, offset0)
            xcall = lambda k: self.lib.bsvi_train_persistent_exchange(
                self.native.handle, C.byref(args), C.byref(cfg), ptr(self.params), ptr(state), ptr(self.mask_all),
                ptr(self.mask_first), int(pretraining_iterations), k, ptr(loss_curve), ptr(finite), ex.handle)
            if xplan is None:
                ok = 1.0 if (ex is not None and xcall(0) == 0) else 0.0
                if world > 1:
                    import torch.distributed as dist
                    vote = torch.tensor([ok], device=dev if dist.get_backend() == "nccl" else "cpu")
                    dist.all_reduce(vote, op=dist.ReduceOp.MIN)
                    ok = float(vote.item())
                xplan = self._train_plans[xkey] = ok > 0.0
            if xplan and ex is not None:
                native.check(xcall(K))
                self.last_mode = "persistent+exchange"
                check_exchange(dev, self.params)
                return loss_curve, finite
        if sharded and noise_t is None and os.environ.get("BSVI_GRAPH", "1") != "0" and not self._externals and not self._minibatches:
            # multi-GPU: the step sequence is captured once in a HIP graph and replayed — no Python between the launches
            try:
                self._train_graph(K, n_local, number_samples, base, cfg, state, loss_curve, finite, seed, offset0,
                                  int(pretraining_iterations))
                self.last_mode = "graph" if world == 1 else "graph+allreduce"
                check_exchange(dev, self.params)
                return loss_curve, finite
            except _CaptureRefused as err:      # capture refused on some rank (every rank is here then): launch by launch below
                warnings.warn("HIP-graph capture of the sharded step failed ({}); stepping eagerly".format(err))
                loss_curve.zero_()
        for it in range(K):
            nz = None if noise_t is None else noise_t[it]
            if self._externals:
                nz = self._external_rows(n_local, number_samples, base, nz, seed, offset0 + it,
                                         None if minibatch_seq is None else minibatch_seq[it])
            if self._minibatches:
                self._refresh_minibatches(seed, offset0 + it, None if minibatch_seq is None else minibatch_seq[it])
            args = self._elbo_args(n_local, number_samples, base, nz, seed, offset0 + it)
            mask = self.mask_all if it > pretraining_iterations else self.mask_first
            if world == 1 and not _force_sharded_path:
                self.native.attach_shares()
                native.check(self.lib.bsvi_svi_step(
                    self.native.handle, C.byref(args), C.byref(cfg), ptr(self.params), ptr(state), ptr(mask),
                    C.c_void_p(loss_curve.data_ptr() + 4 * it), C.c_void_p(finite.data_ptr() + 4 * it)))
            else:
                self.native.attach_shares()
                native.check(self.lib.bsvi_elbo_fwd_bwd(self.native.handle, C.byref(args)))
                allreduce_sums(self.out)
                native.check(self.lib.bsvi_finalize_step(
                    C.byref(cfg), ptr(self.params), ptr(self.out), ptr(state), ptr(mask), p.n_params, number_samples,
                    C.c_void_p(loss_curve.data_ptr() + 4 * it), C.c_void_p(finite.data_ptr() + 4 * it), self._stream()))
        self.last_mode = "stepwise" if world == 1 else "stepwise+allreduce"
        if world > 1:
            check_exchange(dev, self.params)
        return loss_curve, finite


def training_buffers(n_iterations, n_params, device, with_state=True):
    """loss curve [K], finite flags [K] and optimizer state [4 P] of one `train` call as views of ONE allocation: a single
    fill launch in front of the training launch instead of three (every kernel path writes the loss and the flag of each
    iteration it runs, so their initial value is never read; the optimizer state must start at 0).  with_state=False: the
    kernel keeps the optimizer state to itself — nothing to fill, no launch at all."""
    K, P = max(int(n_iterations), 1), max(int(n_params), 1)
    Ka = (K + 3) // 4 * 4                               # views stay 16-byte aligned
    if not with_state:
        buf = torch.empty(2 * Ka, device=device)
        return buf[:K], buf[Ka:Ka + K], None
    buf = torch.zeros(2 * Ka + 4 * P, device=device)
    return buf[:K], buf[Ka:Ka + K], buf[2 * Ka:]


def _graph_unroll():
    return max(1, int(os.environ.get("BSVI_GRAPH_UNROLL", "32")))


def compile_model(joint_model, posterior_model=None, gradient_estimator=None):
    """Compile once per (posterior, estimator); cached on the joint model."""
    if posterior_model is None:
        joint_model.check_posterior_model()
        posterior_model = joint_model.posterior_model
    est = estimator_name(gradient_estimator)
    key = (id(posterior_model), est)
    compiled = joint_model._compiled.get(key)
    if compiled is None:
        # all programs of one (joint, posterior) pair share one parameter buffer
        sibling = next((c for (pid, _), c in joint_model._compiled.items() if pid == id(posterior_model)), None)
        try:
            compiled = CompiledELBO(joint_model, posterior_model, est)
        except lowering.LoweringError as scalar_error:
            # graphs with a dense matmul link go to the MFMA path (dense.py); anything else is an error
            from brancher_amd import dense
            try:
                compiled = dense.CompiledDense(joint_model, posterior_model, est)
            except lowering.LoweringError as dense_error:
                # a chain of matmul links with latent matrices AND biases — a Bayesian neural network — goes to bnn.py
                from brancher_amd import bnn
                try:
                    compiled = bnn.CompiledBnn(joint_model, posterior_model, est)
                except lowering.LoweringError as bnn_error:
                    # encoder / decoder network links go to the amortised path (amortized.py)
                    from brancher_amd import amortized
                    try:
                        compiled = amortized.CompiledAmortized(joint_model, posterior_model, est)
                    except lowering.LoweringError as amort_error:
                        raise lowering.LoweringError("{}; dense path: {}; bnn path: {}; amortised path: {}".format(
                            scalar_error, dense_error, bnn_error, amort_error)) from None
        if sibling is not None:
            if sibling.n_params != compiled.n_params:
                raise RuntimeError("parameter layouts of two estimators of the same model differ")
            compiled.params = sibling.params
            for par, off, size, _ in compiled.program.parameters:
                par.bind(compiled, off)
        joint_model._compiled[key] = compiled
    return compiled


def is_custom_estimator(gradient_estimator):
    """a user-defined `GradientEstimator` subclass (seam B2, `gradient_estimators.py:17-26`): a class (or instance) with its
    own `__call__` and no kernel of its own"""
    from brancher_amd import gradient_estimators as ge
    cls = gradient_estimator if isinstance(gradient_estimator, type) else type(gradient_estimator)
    return gradient_estimator is not None and not isinstance(gradient_estimator, str) \
        and isinstance(cls, type) and issubclass(cls, ge.GradientEstimator) and getattr(cls, "kernel_name", None) is None


class _OpaqueSamples(dict):
    """what `sampler._get_sample` hands a user-defined estimator: the draw itself stays on the device inside the kernel;
    the estimator passes it on to `function` / `calculate_log_probability` unchanged (`samples.update(empirical)` is fine)"""


def custom_estimator_loss(joint_model, posterior_model, estimator_cls, number_samples, noise=None, minibatch=None):
    """A user-defined gradient estimator (`gradient_estimators.py:17-26`, instantiated at `variables.py:856-857`): a torch
    scalar g built from `self.function(samples)` — f_n = log p + entropy per sample — and
    `self.sampler.calculate_log_probability(samples)` — log q_n — of ONE draw `self.sampler._get_sample(n)`.
    Two passes of the fused kernel around the user's torch code: (1) f and log q per sample (`fvalue_out_dev`); the
    estimator runs on two leaf tensors holding them, and autograd on those N-vectors gives a_n = dg/df_n, b_n = dg/dlog q_n;
    (2) the same draw again with the weights: the output block receives d(-g)/d theta = -(sum_n a_n grad f_n + b_n grad log q_n),
    reparameterisation path included (the reference drops `differentiable=False`, `variables.py:567`).  Returns +g as a
    FusedLoss.  The estimator must hand the sample object on untouched — a Taylor1-style substitution of values is a
    different program (and built in).  Served on all three paths: the scalar path, the dense-link path (whose gradient of
    log q_n is -1 / scale for every sample, so b enters as its sum: dense_kernel.inc) and the amortised path (one weight per
    (sample, minibatch row): the row kernels scale the gradient seeds, every launch behind them is linear in the seeds);
    `minibatch`: the rows of the model's minibatch variable (parity tests), else both passes draw the same rows from
    (seed, offset)."""
    if isinstance(estimator_cls, type):
        make = estimator_cls
    else:
        make = type(estimator_cls)
    compiled = compile_model(joint_model, posterior_model, "blackbox")
    kind = type(compiled).__name__
    if kind not in ("CompiledELBO", "CompiledDense", "CompiledBnn", "CompiledAmortized"):
        raise NotImplementedError("user-defined gradient estimators: unknown engine " + kind)
    extra = dict(minibatch=minibatch) if (kind != "CompiledELBO" or getattr(compiled, "_minibatches", None)
                                          or any(getattr(n.node, "kind", "mvn") == "reduce" for n in getattr(compiled, "_externals", []))) else {}
    N = int(number_samples)
    seed = compiled._seed(None)
    offset = compiled.iteration
    compiled.iteration += 1
    rank, world = dist_info()
    if world > 1:
        raise NotImplementedError("user-defined gradient estimators on several ranks")
    first = compiled.evaluate(N, seed=seed, offset=offset, want_fvalues=True, noise=noise, **extra)     # (noise: parity tests)
    # (the amortised path: one value per (sample, minibatch row), the reference's [number_samples, batch_size])
    shape = (N, compiled.program.batch_size) if kind == "CompiledAmortized" else (-1, 1)
    F = first["f"].detach().clone().reshape(shape).requires_grad_(True)
    lq = first.get("lq")               # (a dense-link point estimate has no q: log q = 0, as the reference's RootVariable gives)
    LQ = (lq if lq is not None else torch.zeros_like(first["f"])).detach().clone().reshape(shape).requires_grad_(True)
    token = _OpaqueSamples()
    drawn = []

    def check(samples):
        if samples is not token:
            raise NotImplementedError("a user-defined gradient estimator must pass the samples of sampler._get_sample on "
                                      "unchanged (the draw lives in the kernel)")

    class Sampler:
        variables = getattr(posterior_model, "variables", [])

        def _get_sample(self, n_samples, differentiable=True, **_):
            if int(n_samples) != N or drawn:
                raise NotImplementedError("a user-defined gradient estimator draws ONE sample set of number_samples")
            drawn.append(True)
            return token

        def calculate_log_probability(self, samples, **_):
            check(samples)
            return LQ

    def function(samples):
        check(samples)
        return F

    value = make(function, Sampler(), {})(N)
    if not torch.is_tensor(value) or value.numel() != 1:
        raise ValueError("a gradient estimator returns a scalar tensor")
    if value.requires_grad:
        value.backward()
    a = F.grad if F.grad is not None else torch.zeros_like(F)
    b = LQ.grad if LQ.grad is not None else torch.zeros_like(LQ)
    compiled.evaluate_weighted(N, a, b, seed, offset, noise=noise, **extra)
    return FusedLoss(compiled, value.detach().reshape(()), grad_scale=-1.0)


def estimate_elbo(joint_model, posterior_model, number_samples, for_gradient=False, gradient_estimator=None):
    if for_gradient and is_custom_estimator(gradient_estimator):
        return custom_estimator_loss(joint_model, posterior_model, gradient_estimator, number_samples)
    if is_custom_estimator(gradient_estimator):
        gradient_estimator = None           # (the value alone: mean of log p + entropy, `variables.py:858-866`)
    compiled = compile_model(joint_model, posterior_model, gradient_estimator)
    res = compiled.evaluate(number_samples)
    if for_gradient:
        # estimate_log_model_evidence returns +ELBO = -loss; the output block holds d loss / d theta
        return FusedLoss(compiled, -res["loss"], grad_scale=-1.0)
    return -res["loss"]


# ---- sampling / evaluation API edge ------------------------------------------------
_sample_tick = [0]


def _run_sampler(model, posterior_model, number_samples, input_values):
    """Ancestral sampling through the same fused kernel (a SAMPLE-only program).  Returns
    {variable: device tensor in the reference layout [N, B, d1, d2]} for every variable sampled,
    plus tiled roots (`RootVariable._get_sample`, variables.py:367-375)."""
    from brancher_amd.variables import RootVariable
    program = lowering.lower_sampler(model, posterior_model, input_values)
    run = CompiledELBO(model, posterior_model, program=program, bind_parameters=False)
    _sample_tick[0] += 1
    # (sampling programs draw at offsets from 2^62: bits 8..21 of the high word carry the retry index of a rejection
    #  sampler (philox_raw), so an offset base of 2^40 made attempt 1 of a sampler call reuse the stream of a training
    #  iteration's attempt 0)
    res = run.evaluate(number_samples, want_samples=True, offset=(1 << 62) + _sample_tick[0])
    samples = res["samples"]
    out = {}
    for var, slot in program.outputs:
        rows = samples[slot.base:slot.base + slot.size]
        out[var] = rows.t().reshape((number_samples,) + tuple(slot.shape)).contiguous()
    models = [model] + ([posterior_model] if posterior_model is not None else [])
    for m in models:
        for var in m.flatten():
            if isinstance(var, RootVariable) and var not in out:
                value = var.value
                if not isinstance(value, np.ndarray):
                    continue
                t = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32)).to(run.device)
                out[var] = t.expand((number_samples,) + tuple(t.shape[1:])).contiguous() if t.shape[0] == 1 else t
    for var, value in (input_values or {}).items():
        out[var] = value if torch.is_tensor(value) else torch.as_tensor(np.asarray(value, dtype=np.float32))
    return out


def _amortized_engine(model):
    """the amortised engine of a joint model whose links are encoder / decoder networks (amortized.py), or None"""
    posterior = getattr(model, "posterior_model", None)
    if not posterior:
        return None
    compiled = compile_model(model, posterior, None)
    return compiled if type(compiled).__name__ == "CompiledAmortized" else None


def _given(input_values, var):
    """the caller's value for `var`: keyed by the variable or by its name"""
    for key, value in (input_values or {}).items():
        if key is var or key == var.name or getattr(key, "name", None) == var.name:
            return value.detach().cpu().numpy() if torch.is_tensor(value) else np.asarray(value, dtype=np.float32)
    return None


def _amortized_sample(model, compiled, number_samples, input_values):
    """`ProbabilisticModel._get_sample` of an amortised model (variables.py:732-742; the posterior-predictive step of
    examples/VAE_playground.py:90-103): z from its prior or from the caller, the decoder's outputs through
    bsvi_amort_apply, x from its likelihood.  Device tensors in the reference's layout: z [N, 1, Dz],
    decoder_output {key: [N, 1, P]}, x [N, 1, P, 1]."""
    from brancher_amd.variables import RootVariable
    p, dev, N = compiled.program, compiled.device, int(number_samples)
    by_name = {v.name: v for v in model.flatten()}
    z_var, x_var = by_name[p.latent_name], by_name[p.data_name]
    _sample_tick[0] += 1
    gen = torch.Generator(device=dev)
    gen.manual_seed((1 << 40) + _sample_tick[0])
    named = compiled.named_params()
    given_z = _given(input_values, z_var)
    if given_z is not None:
        rows = given_z.reshape(-1, p.latent_dim)[:N] if given_z.size >= N * p.latent_dim else given_z.reshape(1, p.latent_dim)
        z = torch.from_numpy(np.array(np.broadcast_to(rows, (N, p.latent_dim)), dtype=np.float32)).to(dev)
    else:
        loc, scale = p.prior_loc, p.prior_scale
        for par, off, size, _ in p.parameters:               # a learnable prior: the current values
            if off == p.prior_loc_off:
                loc = named[par.name].reshape(-1)
            if off == p.prior_scale_off:
                raw = named[par.name].reshape(-1).astype(np.float64)
                scale = np.where(raw > 20, raw, np.log1p(np.exp(np.minimum(raw, 20)))).astype(np.float32)
        z = torch.from_numpy(np.asarray(loc, dtype=np.float32)).to(dev) + \
            torch.from_numpy(np.asarray(scale, dtype=np.float32)).to(dev) * torch.randn((N, p.latent_dim), device=dev, generator=gen)
    out = {z_var: z.reshape(N, 1, p.latent_dim)}
    decoded = {key: compiled.decode(z, key) for key in p.dec_outputs}
    for var in model.flatten():
        if getattr(var, "_type", None) == "Deterministic node" and var.name not in (p.latent_name, p.data_name):
            out[var] = {key: value.reshape(N, 1, -1) for key, value in decoded.items()}
    given_x = _given(input_values, x_var)
    if given_x is not None:
        x = torch.from_numpy(np.array(np.broadcast_to(given_x.reshape(-1, p.n_features)[:N], (N, p.n_features)),
                                      dtype=np.float32)).to(dev)
    elif p.likelihood == "normal":
        if getattr(p, "dec_scale_key", None) is not None:      # a second head of the decoder
            lik_scale = decoded[p.dec_scale_key]
        elif getattr(p, "lik_scale_size", 0):     # learnable: softplus of the raw values the optimizer steps (one, or one per feature)
            raw = compiled.params[p.lik_scale_off:p.lik_scale_off + p.lik_scale_size]
            lik_scale = torch.nn.functional.softplus(raw).expand(p.n_features) if p.lik_scale_size == 1 else torch.nn.functional.softplus(raw)
        else:
            lik_scale = torch.from_numpy(p.likelihood_scale).to(dev)
        x = decoded[p.logits_key] + lik_scale * torch.randn((N, p.n_features), device=dev, generator=gen)
    else:
        x = torch.bernoulli(torch.sigmoid(decoded[p.logits_key]), generator=gen)
    out[x_var] = x.reshape(N, 1, p.n_features, 1)
    for var in model.flatten():                                # tiled roots (`RootVariable._get_sample`, variables.py:367-375)
        if isinstance(var, RootVariable) and var not in out and isinstance(var.value, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(var.value, dtype=np.float32)).to(dev)
            out[var] = t.expand((N,) + tuple(t.shape[1:])).contiguous() if t.shape[0] == 1 else t
    return out


def _amortized_posterior_sample(model, compiled, number_samples):
    """`ProbabilisticModel._get_posterior_sample` of an amortised model (variables.py:796-805, 903-907): one minibatch per
    sample and z = loc(x) + scale(x) eps from the encoder, keyed by the JOINT model's variables of the same names; the
    joint model then has nothing left to sample (both of its random variables are given) — x [N, B, P, 1], z [N, B, Dz]."""
    p, N = compiled.program, int(number_samples)
    by_name = {v.name: v for v in model.flatten()}
    keep = compiled.out.clone()                       # (the draw runs through bsvi_amort_fwd_bwd: leave the caller's gradients alone)
    _sample_tick[0] += 1
    res = compiled.evaluate(N, seed=None, offset=(1 << 62) + _sample_tick[0], want_noise=True, want_indices=True)
    idx, eps = res["indices"].long().reshape(-1), res["noise"]
    compiled.out.copy_(keep)
    rows = torch.from_numpy(p.dataset).to(compiled.device)[idx]
    z = compiled.encode(rows, p.loc_key) + compiled.encode(rows, p.scale_key) * eps
    return {by_name[p.data_name]: rows.reshape(N, p.batch_size, p.n_features, 1),
            by_name[p.latent_name]: z.reshape(N, p.batch_size, p.latent_dim)}


def sample_model(model, number_samples, observed=False, input_values={}):
    """`ProbabilisticModel._get_sample` (variables.py:732-742)."""
    if observed:
        from brancher_amd.variables import RandomVariable
        return {v: torch.from_numpy(v._observed_value) for v in model._flatten()
                if isinstance(v, RandomVariable) and v.is_observed and v.has_observed_value}
    try:
        return _run_sampler(model, None, number_samples, input_values)
    except lowering.LoweringError as scalar_error:
        try:
            compiled = _amortized_engine(model)
        except lowering.LoweringError:
            compiled = None
        if compiled is None:
            raise scalar_error
        return _amortized_sample(model, compiled, number_samples, input_values)


def sample_variables(variables, number_samples, observed=False, input_values={}):
    """`Variable._get_sample` (variables.py:367-375, 527-570): the variable and its ancestors."""
    from brancher_amd.variables import ProbabilisticModel
    return sample_model(ProbabilisticModel(list(variables)), number_samples, observed, input_values)


def posterior_sample(model, number_samples, input_values={}):
    """`ProbabilisticModel._get_posterior_sample` (variables.py:796-805): posterior draws re-keyed to the
    joint model's variables by name, then the remaining variables of the joint model (posterior predictive)."""
    try:
        raw = _run_sampler(model, model.posterior_model, number_samples, input_values)
    except lowering.LoweringError as scalar_error:
        try:
            compiled = _amortized_engine(model)
        except lowering.LoweringError:
            compiled = None
        if compiled is None:
            raise scalar_error
        out = _amortized_posterior_sample(model, compiled, number_samples)
        for var, value in (input_values or {}).items():
            out[var] = value if torch.is_tensor(value) else torch.as_tensor(np.asarray(value, dtype=np.float32))
        return out
    mapping = lowering.get_model_mapping(model.posterior_model, model)
    out = {}
    for var, value in raw.items():
        out[mapping.get(var, var)] = value
    return out


def _frame(sample):
    from brancher_amd.pandas_interface import reformat_sample_to_pandas
    return reformat_sample_to_pandas(sample)


def get_sample_frame(model, number_samples, input_values={}):
    """`get_sample` (variables.py:164-173, 751-757): a DataFrame, one row per sample."""
    from brancher_amd.variables import Variable
    if isinstance(model, Variable):
        return _frame({model: sample_variables([model], number_samples, input_values=input_values)[model]})
    return _frame(sample_model(model, number_samples, input_values=input_values))


def get_posterior_sample_frame(model, number_samples, input_values={}):
    return _frame(posterior_sample(model, number_samples, input_values))


def importance_log_weights(joint_model, posterior_model, q_samples):
    """Per-sample log p(z, y) - log q(z) for caller-supplied posterior samples z — the two terms of
    `ProbabilisticModel.get_importance_weights` (variables.py:821-841).  `q_samples` maps variables (or
    names) to arrays in the reference layout [N, B, d...]; returns device tensors (log_p, log_q) of length N.
    Served by the same fused kernel: an evaluation program whose posterior nodes take their values from the
    caller (BSVI_F_GIVEN) and accumulate log q."""
    named = {}
    n = None
    for var, value in q_samples.items():
        arr = value.detach().cpu().numpy() if hasattr(value, "detach") else np.asarray(value)
        named[getattr(var, "name", var)] = arr
        if arr.ndim >= 1 and arr.shape[0] > 1:
            n = arr.shape[0] if n is None else n
    n = n or 1
    compiled = compile_model(joint_model, posterior_model, "importance")
    noise = noise_from_named(compiled.program, named, n)
    res = compiled.evaluate(n, noise=noise, want_fvalues=True, offset=0)
    return res["f"], res["lq"]


def _variable_log_probability(variables, values, include_parents, reevaluate=True):
    """`Variable.calculate_log_probability(values, reevaluate=…, include_parents=…)` (variables.py:486-520): the variable's own
    log-probability at the caller's values plus that of its ancestors.  Served by the evaluation program of
    `importance_log_weights` on the variable's own graph: the model is ProbabilisticModel([var]) (the variable and its
    ancestors), its "posterior" the unobserved random variables among them, which take the caller's values
    (BSVI_F_GIVEN); observed variables use their data.  An own term is the difference between a graph with and without
    its variable.  As in the reference, with `reevaluate=True` (the default of the variable-level call) the recursion over
    `parents` does NOT stop at a visited node: an ancestor reached along k paths is counted k times; `reevaluate=False`
    counts every ancestor once (what the model-level call does).  Device tensor [N, 1]."""
    from brancher_amd.variables import ProbabilisticModel, RandomVariable

    def is_random(v):
        return isinstance(v, RandomVariable) and getattr(v, "_type", None) != "Deterministic node"

    def graph_log_p(members):
        graph = ProbabilisticModel(list(members))
        latents = [v for v in graph.flatten() if is_random(v) and not v.is_observed]
        given = {}
        for v in latents:
            value = _given(values, v)
            if value is None:
                raise ValueError("calculate_log_probability needs a value for {!r}".format(v.name))
            given[v] = value
        log_p, _ = importance_log_weights(graph, ProbabilisticModel(latents), given)
        return log_p.reshape(-1, 1)

    def random_parents(v):          # the random variables directly above v (looking through deterministic nodes and roots)
        found, stack = [], list(v.parents)
        while stack:
            p = stack.pop()
            if is_random(p):
                if p not in found:
                    found.append(p)
            else:
                stack.extend(p.parents)
        return found

    def own(v):
        above = random_parents(v)
        return graph_log_p([v]) - graph_log_p(above) if above else graph_log_p([v])

    (var,) = variables
    if not include_parents:
        return own(var)
    total = graph_log_p([var])
    if not reevaluate:
        return total
    paths = {}                      # ancestor -> number of paths from var (the reference's recursion visits it that often)

    def walk(v, weight):
        for p in v.parents:
            if is_random(p):
                paths[p] = paths.get(p, 0) + weight
            walk(p, weight)

    walk(var, 1)
    for ancestor, count in sorted(paths.items(), key=lambda kv: kv[0].name):
        if count > 1:
            total = total + (count - 1) * own(ancestor)
    return total


def log_probability(variables, values, include_parents=True, model=None, reevaluate=True):
    """`ProbabilisticModel.calculate_log_probability(rv_values)` (variables.py:718-727): the sum of the node
    log-probabilities of a model at caller-supplied values of its latent variables (observed variables use their
    observed values), one number per sample — a device tensor [N, 1].  Served by the evaluation program of
    `importance_log_weights`: for a joint model it is the `log p` output, for its posterior model the `log q` output."""
    if model is None:
        return _variable_log_probability(list(variables), values, include_parents, reevaluate)
    joint = getattr(model, "joint_model", None)
    if joint is not None:                                   # a PosteriorModel: log q(values)
        _, log_q = importance_log_weights(joint, model, values)
        return log_q.reshape(-1, 1)
    posterior = model.posterior_model
    if not posterior:
        raise NotImplementedError("calculate_log_probability needs the model's posterior to be set: the values are "
                                  "matched to the model's latent variables by name through it")
    log_p, _ = importance_log_weights(model, posterior, values)
    return log_p.reshape(-1, 1)
