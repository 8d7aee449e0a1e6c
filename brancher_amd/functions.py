"""
Symbolic function namespace (``import brancher_amd.functions as BF``).

The reference wraps *every* public function of ``torch.nn.functional`` and
``torch._C._VariableFunctions`` into a ``BrancherFunction`` at import time
(`brancher/functions.py:50-62`).  Here ``BF.<name>(...)`` builds a ``call`` node of the
link expression DAG for any name (PEP 562 module ``__getattr__``); whether the fused
kernel can execute it is decided by the lowering, which supports the closed set the
reference's examples/tests actually use (SURVEY §2 census: matmul, exp, delta, sin, tanh,
sum, softplus, sigmoid, sqrt, cos, abs, ...) and raises ``NotImplementedError`` with the
offending name otherwise.
"""
from brancher_amd.variables import var2link, Variable, PartialLink
from brancher_amd import symbolic as sym


def _is_torch_module(fn):
    try:
        import torch
    except ImportError:  # pragma: no cover
        return False
    return isinstance(fn, torch.nn.Module)


class ModuleLink:
    """A ``torch.nn.Module`` used as a link (`functions.py:15-20`, `examples/VAE_playground.py:66-67`).

    The reference calls the module on every ELBO evaluation and lets ``torch.optim`` update its
    ``nn.Parameter``s.  Here the module is a *description*: `amortized.trace_network` reads its layer
    structure once (torch.fx), its tensors are copied into ``Parameter`` objects (segments of the engine's
    flat HBM buffer, torch layout kept) and the MFMA kernels of `csrc/amort_kernel.hip` do the arithmetic.
    ``sync_to_module()`` writes the trained values back into the module."""
    _count = 0

    def __init__(self, module, name):
        from brancher_amd.modules import Parameter
        self.module = module
        ModuleLink._count += 1
        self.name = "{}#{}".format(name, ModuleLink._count)
        # (the tensors are named after the function's name when it was given one — `BrancherFunction(net, name="net")` ->
        #  "net.0.weight" — so that parameters and gradients can be addressed by name; anonymous functions keep the counter)
        prefix = self.name if "?" in name else name
        self.named = {pname: Parameter(p.detach().cpu().numpy(), name="{}.{}".format(prefix, pname))
                      for pname, p in module.named_parameters()}

    def parameters(self):
        return list(self.named.values())

    def sync_to_module(self):
        import torch
        with torch.no_grad():
            for pname, p in self.module.named_parameters():
                p.copy_(torch.from_numpy(self.named[pname].numpy().copy()).to(p.device))

    def __call__(self, *args, **kwargs):
        raise RuntimeError("a ModuleLink is evaluated by the native engine, not called")


class BrancherFunction(object):
    """Lifts a backend function (by name) or a user callable/module to symbolic links
    (`brancher/functions.py:9-45`)."""

    def __init__(self, fn, name="f_?"):
        self.fn = fn
        self.name = name if isinstance(fn, str) or name != "f_?" else getattr(fn, "__name__", name)
        self.links = set()
        if not isinstance(fn, str) and hasattr(fn, "parameters") and callable(getattr(fn, "parameters")):
            # an optimizable module (`functions.py:15-20`).  A torch.nn.Module is held through a ModuleLink: its
            # tensors become segments of the engine's flat parameter buffer, the module itself is never called
            if _is_torch_module(fn):
                fn = self.fn = ModuleLink(fn, self.name)
            self.links = {fn}

    def _get_string(self, *args, **kwargs):
        def s(a):
            l = var2link(a)
            return l.string if isinstance(l, PartialLink) else str(a)
        return self.name + "(" + ", ".join([s(a) for a in list(args) + list(kwargs.values())]) + ")"

    def _traced(self, args, kwargs):
        """A plain Python callable (`BrancherFunction(lambda a, b: torch.exp(a) * 0.5 + torch.tanh(b))`,
        `functions.py:28-41`): called ONCE with the symbolic arguments — arithmetic on links builds links, torch functions
        dispatch through ``BrancherClass.__torch_function__`` — so that the closure becomes an ordinary link expression
        the lowering can compile.  None when the callable does something that cannot be traced (it then stays an
        opaque node and the lowering names it)."""
        if isinstance(self.fn, str) or self.links or not callable(self.fn):
            return None
        try:
            out = self.fn(*[var2link(a) if isinstance(a, (Variable, PartialLink)) else a for a in args],
                          **{k: var2link(v) if isinstance(v, (Variable, PartialLink)) else v for k, v in kwargs.items()})
        except Exception:
            return None
        if isinstance(out, Variable):
            out = var2link(out)
        if isinstance(out, PartialLink):
            return PartialLink(out.vars, out.expr, out.links, string=self._get_string(*args, **kwargs))
        return None

    def __call__(self, *args, **kwargs):
        traced = self._traced(args, kwargs)
        if traced is not None:
            return traced
        link_args = [var2link(arg) for arg in args]
        link_kwargs = {name: var2link(arg) for name, arg in kwargs.items()}
        vars_ = set()
        links = set(self.links)
        for l in list(link_args) + list(link_kwargs.values()):
            if isinstance(l, PartialLink):
                vars_ |= l.vars
                links |= l.links
        e_args = [l.expr if isinstance(l, PartialLink) else l for l in link_args]
        e_kwargs = {k: (l.expr if isinstance(l, PartialLink) else l) for k, l in link_kwargs.items()}
        return PartialLink(vars_, sym.call(self.fn, e_args, e_kwargs), links,
                           string=self._get_string(*args, **kwargs))


# custom functions of the reference (`functions.py:65-66`, `utilities.py:341-358`)
batch_meshgrid = BrancherFunction("batch_meshgrid", "batch_meshgrid")
delta = BrancherFunction("delta", "delta")

_cache = {}


def __getattr__(name):
    if name.startswith("_"):
        raise AttributeError(name)
    if name not in _cache:
        _cache[name] = BrancherFunction(name, name)
    return _cache[name]
