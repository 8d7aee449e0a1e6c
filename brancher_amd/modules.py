"""
Parameter containers.

Counterpart of `brancher/modules.py:9-26` (``ParameterModule`` holds one learnable
tensor, ``EmptyModule`` is the registry the optimizer fills).  The reference keeps each
parameter as its own ``nn.Parameter``; here every learnable value is a *segment of one
flat fp32 buffer in HBM* that the fused ELBO kernel reads and the fused optimizer kernel
updates in place.  Before a model is compiled the value lives in host memory (numpy);
afterwards ``Parameter.data`` is a view onto the engine's flat buffer.
"""
import numpy as np


class Parameter:
    """One learnable tensor (stored unconstrained, `standard_variables.py:66`)."""

    def __init__(self, data, name=None):
        self._host = np.ascontiguousarray(np.asarray(data, dtype=np.float32))
        self.shape = self._host.shape
        self.name = name
        self._store = None      # engine-side flat buffer (ParameterStore) once bound
        self._offset = None
        self.requires_grad = True

    # -- binding to a device-resident flat buffer ------------------------------------
    def bind(self, store, offset):
        self._store = store
        self._offset = int(offset)

    def unbind(self):
        if self._store is not None:
            self._host = self.numpy().copy()
        self._store = None
        self._offset = None

    @property
    def size(self):
        return int(np.prod(self.shape)) if len(self.shape) else 1

    def numpy(self):
        if self._store is None:
            return self._host
        return self._store.read_params(self._offset, self.size).reshape(self.shape)

    def set(self, value):
        value = np.ascontiguousarray(np.asarray(value, dtype=np.float32)).reshape(self.shape)
        if self._store is None:
            self._host = value.copy()
        else:
            self._store.write_params(self._offset, value.reshape(-1))

    @property
    def data(self):
        return self.numpy()

    @property
    def grad(self):
        if self._store is None:
            return None
        g = self._store.read_grads(self._offset, self.size)
        return None if g is None else g.reshape(self.shape)

    def __repr__(self):
        return "Parameter(name=%r, shape=%s)" % (self.name, tuple(self.shape))


class ParameterModule:
    """Holds one ``Parameter``; calling it returns the parameter (`modules.py:9-18`)."""

    def __init__(self, parameter):
        self.parameter = parameter

    def __call__(self, *args, **kwargs):
        return self.parameter

    def parameters(self):
        return [self.parameter]


class EmptyModule(list):
    """Ordered registry of parameter holders (`modules.py:20-26`)."""

    def parameters(self):
        seen, out = set(), []
        for link in self:
            for p in link.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        return out
