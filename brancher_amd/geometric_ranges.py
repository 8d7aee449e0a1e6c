"""
Constraint bijectors for auto-created parameters (`brancher/geometric_ranges.py`).

Learnable values are stored unconstrained; the forward transform is part of the link
expression and therefore runs inside the fused kernel (for the standard cases it is
hoisted into the kernel's lane-uniform table).
"""
from abc import ABC, abstractmethod

import numpy as np

import brancher_amd.functions as BF


class GeometricRange(ABC):

    @abstractmethod
    def forward_transform(self, x, dim):
        pass

    @abstractmethod
    def inverse_transform(self, x, dim):
        pass


class UnboundedRange(GeometricRange):

    def forward_transform(self, x, dim):
        return x

    def inverse_transform(self, y, dim):
        return y


class Interval(GeometricRange):
    # `geometric_ranges.py:34-45`

    def __init__(self, lower_bound, upper_bound):
        self.lower_bound = lower_bound
        self.upper_bound = upper_bound

    def forward_transform(self, x, dim):
        return self.lower_bound + (self.upper_bound - self.lower_bound) * BF.sigmoid(x)

    def inverse_transform(self, y, dim):
        z = (y - self.lower_bound) / (self.upper_bound - self.lower_bound)
        return np.log(z / (1 - z))


class RightHalfLine(GeometricRange):
    # `geometric_ranges.py:48-57`

    def __init__(self, lower_bound):
        self.lower_bound = lower_bound

    def forward_transform(self, x, dim):
        return self.lower_bound + BF.softplus(x)

    def inverse_transform(self, y, dim):
        return np.log(np.exp(y - self.lower_bound) - 1)


class LeftHalfLine(GeometricRange):
    # `geometric_ranges.py:60-69` (the attribute is called lower_bound there too)

    def __init__(self, upper_bound):
        self.lower_bound = upper_bound

    def forward_transform(self, x, dim):
        return self.lower_bound - BF.softplus(x)

    def inverse_transform(self, y, dim):
        return np.log(np.exp(-y + self.lower_bound) - 1)


class Simplex(GeometricRange):
    # `geometric_ranges.py:72-81`

    def forward_transform(self, x, dim):
        latent_p = BF.softplus(x)
        normalization = BF.sum(latent_p, axis=1, keepdims=True)
        normalization = BF.broadcast_to(normalization, latent_p.shape())
        return latent_p / normalization

    def inverse_transform(self, y, dim):
        return np.log(np.exp(y) - 1)


class PositiveDefiniteMatrix(GeometricRange):
    # `geometric_ranges.py:84-91`

    def forward_transform(self, x, dim):
        return BF.matmul(x, BF.transpose(x, -2, -1))

    def inverse_transform(self, y, dim):
        return np.linalg.cholesky(y)
