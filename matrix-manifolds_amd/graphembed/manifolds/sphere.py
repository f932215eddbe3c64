"""Unit sphere — counterpart of graphembed/graphembed/manifolds/sphere.py:8-99."""
import numpy as np
import torch

from graphembed import _backend as B
from graphembed.manifolds.base import _like
from graphembed.manifolds.euclidean import _shape_name
from graphembed.manifolds.vector import VectorManifold


class Sphere(VectorManifold):
    _kind = B.SPHERE
    use_gram = True

    def __init__(self, *shape):
        if len(shape) == 0:
            raise ValueError('Need shape parameters.')
        self.shape = shape
        self._name = _shape_name('Sphere', shape)
        self.dims = tuple(np.arange(-len(shape), 0))

    @property
    def dim(self):
        return np.prod(self.shape) - 1

    def zero(self, *shape, out=None):  # sphere.py:30-33: the point (-1, 0, ..., 0)
        x = torch.zeros(*shape, int(np.prod(self.shape)), **_like(out))
        x[..., 0] = -1
        return x.reshape(*shape, *self.shape)

    def zero_vec(self, *shape, out=None):
        return torch.zeros(*shape, *self.shape, **_like(out))

    def inner(self, x, u, v, keepdim=False):
        return (u * v).sum(self.dims, keepdim=keepdim)

    def rand(self, *shape, out=None, ir=1e-2):  # sphere.py:76-79
        x = self.zero(*shape, out=out)
        with torch.no_grad():
            return self.retr(x, self.randvec(x, norm=ir))

    def rand_uniform(self, *shape, out=None):  # sphere.py:81-83
        with torch.no_grad():
            return self.projx(torch.randn(*shape, *self.shape, **_like(out)))

    def rand_ball(self, *shape, out=None):  # sphere.py:85-90
        xs = self.rand_uniform(*shape, out=out)
        rs = torch.rand(*shape, dtype=xs.dtype, device=xs.device).pow_(1 / (self.dim + 1))
        return xs.mul_(rs.reshape(*shape, *((1, ) * len(self.shape))))

    def randvec(self, x, norm=1):  # sphere.py:91-96
        with torch.no_grad():
            u = self.proju(x, torch.randn_like(x))
        return u.div_(u.norm(dim=self.dims, keepdim=True)).mul_(norm)

    def __str__(self):
        return self._name
