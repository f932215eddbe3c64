"""Flat space — counterpart of graphembed/graphembed/manifolds/euclidean.py:7-60."""
import numpy as np
import torch

from graphembed import _backend as B
from graphembed.manifolds.base import _like
from graphembed.manifolds.vector import VectorManifold


def _shape_name(kind, shape):
    if len(shape) == 1:
        return '{} manifold of {}-vectors'.format(kind, *shape)
    if len(shape) == 2:
        return '{} manifold of {}x{} matrices'.format(kind, *shape)
    return '{} manifold of shape '.format(kind) + str(shape) + ' tensors'


class Euclidean(VectorManifold):
    _kind = B.EUCLIDEAN

    def __init__(self, *shape):
        if len(shape) == 0:
            raise ValueError('Need shape parameters.')
        self.shape = shape
        self._name = _shape_name('Euclidean', shape)
        self.dims = tuple(np.arange(-len(shape), 0))

    @property
    def dim(self):
        return np.prod(self.shape)

    def zero(self, *shape, out=None):
        return torch.zeros(*shape, *self.shape, **_like(out))

    def zero_vec(self, *shape, out=None):
        return torch.zeros(*shape, *self.shape, **_like(out))

    def inner(self, x, u, v, keepdim=False):
        return (u * v).sum(self.dims, keepdim=keepdim)

    # the flat maps are identities / sums: no kernel launch needed (euclidean.py:38-48)
    def proju(self, x, u, inplace=False):
        return u

    def projx(self, x, inplace=False):
        return x

    def egrad2rgrad(self, x, u):
        return u

    def exp(self, x, u):
        return x + u

    def retr(self, x, u):
        return x + u

    def log(self, x, y):
        return y - x

    def transp(self, x, y, u):
        return u

    def rand(self, *shape, out=None, ir=1e-2):
        return torch.empty(*shape, *self.shape, **_like(out)).uniform_(-ir, ir)

    def randvec(self, x, norm):
        u = torch.randn_like(x)
        return u.div_(u.norm(dim=self.dims, keepdim=True)).mul_(norm)

    def __str__(self):
        return self._name
