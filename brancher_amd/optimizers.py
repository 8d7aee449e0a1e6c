"""
Optimizer registry (API of `brancher/optimizers.py:19-73`).

The reference walks a model's variables, gathers their ``ParameterModule`` / ``LinkConstructor`` links into an
``EmptyModule`` and hands ``module.parameters()`` to ``getattr(torch.optim, name)(..., **kwargs)``.  Here a
``ProbabilisticOptimizer`` only *records* which ``Parameter`` objects it owns and the validated ``torch.optim``
configuration: the update itself is the fused device optimizer (``bsvi_finalize_step`` / the step fused into the
reduction kernels), element for element torch.optim.SGD / torch.optim.Adam.  `inference.perform_inference` builds one
per model exactly like the reference (`inference.py:77-88`) and uses ``.optimizer`` (None when the model has nothing to
learn) to decide which groups exist.
"""
from brancher_amd import native
from brancher_amd.modules import EmptyModule, ParameterModule
from brancher_amd.standard_variables import LinkConstructor
from brancher_amd.variables import ProbabilisticModel, Variable


def _variables_of(model):
    if isinstance(model, ProbabilisticModel):
        return model.flatten()
    if isinstance(model, Variable):
        return model.ancestors
    raise ValueError("Only brancher variables and iterable of variables can be added to a probabilistic optimizer")


def _parameter_holders(link):
    """the objects of a link that own parameters: the link itself, or the modules a LinkConstructor collected"""
    if isinstance(link, ParameterModule):
        return [link]
    if isinstance(link, LinkConstructor):
        return list(link)
    return []


class ProbabilisticOptimizer:

    def __init__(self, model, optimizer='SGD', **kwargs):
        assert isinstance(optimizer, str), 'Optimizer should be a name of available pytoch optimizers'
        self.optimizer_name, self.kwargs = optimizer, dict(kwargs)
        self.link_set = set()
        self.module = EmptyModule()
        self._device_state = {}     # id(compiled program) -> (optimizer state [4][P], ownership mask [P]) on its device
        models = [model] if isinstance(model, (Variable, ProbabilisticModel)) else list(model)
        for m in models:
            self.add_variable2module(m)
        # validates the configuration against what the fused optimizer implements; None = nothing to optimise
        self.optimizer = native.make_opt_cfg(optimizer, **kwargs) if self.module.parameters() else None

    def add_variable2module(self, model):
        """register the parameter holders of every variable of `model` (`optimizers.py:42-51`)"""
        fresh = {getattr(v, "link", None) for v in _variables_of(model)} - self.link_set - {None}
        fresh = {link for link in fresh if _parameter_holders(link)}
        self.link_set |= fresh
        for link in fresh:
            self.module.extend(_parameter_holders(link))

    def parameters(self):
        return self.module.parameters()

    # ---- the reference's step interface (`optimizers.py:69-73`), for hand-written loops in the style of
    #      `inference.py:95-108`:  loss = method.compute_loss(...); opt.zero_grad(); loss.backward(); opt.update()
    def zero_grad(self):
        """nothing to clear: every fused ELBO evaluation overwrites the gradient block of its compiled program"""

    def update(self):
        """One optimizer step of the parameters this optimizer owns, on the device (`bsvi_optimizer_step`), from the
        gradients of the most recent ``compute_loss`` / ``estimate_log_model_evidence(for_gradient=True)`` of the
        compiled program(s) they live in.  Optimizer state (momentum, Adam moments, step counts) is kept per optimizer,
        as `torch.optim` keeps it per instance."""
        import numpy as np
        import torch
        from brancher_amd import engine
        if self.optimizer is None:
            return
        by_store = {}
        for par in self.parameters():
            store = par._store
            if store is None:
                raise RuntimeError("parameter {!r} is not part of a compiled model yet: evaluate the loss first".format(par.name))
            by_store.setdefault(id(store), (store, []))[1].append(par)
        for store, pars in by_store.values():
            if not getattr(store, "grads_valid", False):
                raise RuntimeError("no gradients: call compute_loss(...).backward() before update()")
            key = id(store)
            if key not in self._device_state:
                mask = np.zeros(max(store.n_params, 1), dtype=np.uint8)
                for par in pars:
                    mask[par._offset:par._offset + par.size] = 1
                self._device_state[key] = (torch.zeros(4 * max(store.n_params, 1), device=store.device),
                                           torch.from_numpy(mask).to(store.device))
            state, mask = self._device_state[key]
            engine.optimizer_step(store, self.optimizer, state, mask)
