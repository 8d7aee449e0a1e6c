"""
Host-side layout helpers.

The layout contract is the reference's (`brancher/utilities.py:223-254`): every value is
fp32 shaped ``[N (Monte-Carlo samples), B (datapoints), d1, d2, ...]``.  A python number
becomes ``[1,1,1,1]``; an unobserved array of shape ``s`` becomes ``[1,1,*s]``; an observed
array of shape ``[B, ...]`` becomes ``[1,B,...]`` padded with trailing ones to 4-D.

On the device the engine does *not* use this layout: it keeps structure-of-arrays
``[element][N]`` with the sample axis fastest so that the 64 lanes of a wavefront touch
consecutive addresses.  These helpers convert at the API edge only.
"""
from functools import reduce

import numpy as np

DISCRETE_TYPES = (list, set, tuple, dict, str)


def is_discrete(data):
    # `brancher/utilities.py:31-32`
    return type(data) in DISCRETE_TYPES


def is_tensor(data):
    try:
        import torch
        if torch.is_tensor(data):
            return True
    except Exception:  # pragma: no cover
        pass
    return isinstance(data, np.ndarray)


def to_numpy(data):
    try:
        import torch
        if torch.is_tensor(data):
            return data.detach().cpu().numpy()
    except Exception:  # pragma: no cover
        pass
    return np.asarray(data)


def coerce_to_dtype(data, is_observed=False):
    """numpy restatement of the shape rule of `brancher/utilities.py:223-254`."""
    try:
        import pandas as pd
        if isinstance(data, pd.DataFrame):
            data = data.values
    except Exception:  # pragma: no cover
        pass
    if is_discrete(data):
        return data
    if isinstance(data, (bool, int, float, np.floating, np.integer)):
        result = np.full((1, 1), float(data), dtype=np.float32)
    else:
        try:
            result = to_numpy(data).astype(np.float32)
        except Exception:
            raise TypeError("Invalid input dtype {} - expected float, integer, np.ndarray, or torch var."
                            .format(type(data)))
        if result.ndim == 0:
            result = result.reshape(1, 1)
    if is_observed:
        result = result[None]
        if result.ndim == 2:
            result = result.reshape(result.shape + (1, 1))
        elif result.ndim == 3:
            result = result.reshape(result.shape + (1,))
    else:
        result = result[None, None]
    return np.ascontiguousarray(result, dtype=np.float32)


def join_dicts_list(dicts_list):
    out = {}
    for d in dicts_list:
        out.update(d)
    return out


def join_sets_list(sets_list):
    if sets_list:
        return reduce(lambda a, b: a.union(b), sets_list)
    return set()


def flatten_list(lst):
    return [item for sub in lst for item in sub]


def canonical_elem_shape(shape_after_n):
    """[B, d1, d2, ...] -> (B, D1, D2): pad with trailing ones (the reference's
    ``uniform_shapes`` appends a trailing axis, `utilities.py:274-279`); ranks above
    three are folded into the last axis (only legal for plain elementwise use)."""
    s = tuple(int(x) for x in shape_after_n)
    if len(s) == 0:
        s = (1,)
    if len(s) > 3:
        s = (s[0], s[1], int(np.prod(s[2:])))
    while len(s) < 3:
        s = s + (1,)
    return s


def broadcast_shapes3(*shapes):
    out = [1, 1, 1]
    for s in shapes:
        for i in range(3):
            if s[i] != 1:
                if out[i] != 1 and out[i] != s[i]:
                    raise ValueError("shapes %r cannot be broadcast" % (shapes,))
                out[i] = s[i]
    return tuple(out)
