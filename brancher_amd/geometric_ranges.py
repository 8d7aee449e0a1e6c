"""
Constraints of auto-created parameters (API of `brancher/geometric_ranges.py`).

A standard variable stores every constant / learnable argument *unconstrained* (`standard_variables.py:57-68`):
``inverse_transform`` maps the user's value to the stored one on the host (numpy, once), ``forward_transform`` is the
symbolic link that maps it back — part of the link expression, so it runs on the device; for the standard cases
``a + b * g(root)`` it is hoisted into the kernel's lane-uniform table (`lowering.match_uniform`).

Each range here is an instance of one of three maps:

    identity                         UnboundedRange
    edge + direction * softplus(u)   RightHalfLine (direction +1), LeftHalfLine (-1)
    edge + width * sigmoid(u)        Interval

plus the two matrix/vector ranges of the reference (Simplex, PositiveDefiniteMatrix), which the fused kernels do not
lower (they raise in the lowering, not here).
"""
import numpy as np

import brancher_amd.functions as BF


def _inverse_softplus(v):
    """u with softplus(u) = v, v > 0   (the reference's log(exp(v) - 1), `geometric_ranges.py:56-57`)"""
    return np.log(np.expm1(v))


def _logit(p):
    return np.log(p / (1 - p))


class GeometricRange:
    """``forward_transform(stored, dim)`` -> symbolic constrained value; ``inverse_transform(value, dim)`` -> stored."""

    def forward_transform(self, x, dim):
        raise NotImplementedError(type(self).__name__)

    def inverse_transform(self, y, dim):
        raise NotImplementedError(type(self).__name__)


class UnboundedRange(GeometricRange):
    forward_transform = inverse_transform = lambda self, value, dim: value


class _HalfLine(GeometricRange):
    direction = +1

    def __init__(self, bound):
        # the reference calls the attribute `lower_bound` on both half lines (`geometric_ranges.py:50,62`)
        self.lower_bound = bound

    def forward_transform(self, x, dim):
        bump = BF.softplus(x)
        return self.lower_bound + bump if self.direction > 0 else self.lower_bound - bump

    def inverse_transform(self, y, dim):
        return _inverse_softplus(self.direction * (y - self.lower_bound))


class RightHalfLine(_HalfLine):
    """values above `lower_bound`: scales, concentrations (`geometric_ranges.py:48-57`)"""

    def __init__(self, lower_bound):
        super().__init__(lower_bound)


class LeftHalfLine(_HalfLine):
    """values below `upper_bound` (`geometric_ranges.py:60-69`)"""
    direction = -1

    def __init__(self, upper_bound):
        super().__init__(upper_bound)


class Interval(GeometricRange):
    """values in (lower_bound, upper_bound): probabilities (`geometric_ranges.py:34-45`)"""

    def __init__(self, lower_bound, upper_bound):
        self.lower_bound, self.upper_bound = lower_bound, upper_bound

    def forward_transform(self, x, dim):
        width = self.upper_bound - self.lower_bound
        return self.lower_bound + width * BF.sigmoid(x)

    def inverse_transform(self, y, dim):
        return _logit((y - self.lower_bound) / (self.upper_bound - self.lower_bound))


class Simplex(GeometricRange):
    """positive vectors normalised over the first data axis (`geometric_ranges.py:72-81`)"""

    def forward_transform(self, x, dim):
        positive = BF.softplus(x)
        total = BF.broadcast_to(BF.sum(positive, axis=1, keepdims=True), positive.shape())
        return positive / total

    def inverse_transform(self, y, dim):
        return _inverse_softplus(y)


class PositiveDefiniteMatrix(GeometricRange):
    """a matrix stored through a factor L with value = L L^T (`geometric_ranges.py:84-91`)"""

    def forward_transform(self, x, dim):
        return BF.matmul(x, BF.transpose(x, -2, -1))

    def inverse_transform(self, y, dim):
        return np.linalg.cholesky(y)
