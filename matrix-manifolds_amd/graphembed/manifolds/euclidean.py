"""Flat space R^shape — counterpart of graphembed/graphembed/manifolds/euclidean.py:7-60.

Every map of the Manifold API is an identity, a sum or a difference here, so none of them launches a
kernel of the library; distances, the pair kernels, the fused objective and the fused RSGD step come
from `VectorManifold` with kind EUCLIDEAN (differences, not a Gram form: no cancellation for close
points)."""
import math

import torch

from graphembed import _backend as B
from graphembed.manifolds.base import _like
from graphembed.manifolds.vector import VectorManifold

_NOUN = {1: '{}-vectors', 2: '{}x{} matrices'}


def _shape_name(kind, shape):
    """'<kind> manifold of 10-vectors' / '... of 3x3 matrices' / '... of shape (2, 3, 4) tensors'."""
    noun = _NOUN[len(shape)].format(*shape) if len(shape) in _NOUN else 'shape {} tensors'.format(shape)
    return '{} manifold of {}'.format(kind, noun)


class Euclidean(VectorManifold):
    _kind = B.EUCLIDEAN

    def __init__(self, *shape):
        if not shape:
            raise ValueError('Need shape parameters.')
        self.shape = shape
        self.dims = tuple(range(-len(shape), 0))   # the point dimensions, for reductions
        self._name = _shape_name('Euclidean', shape)

    def __str__(self):
        return self._name

    @property
    def dim(self):
        return math.prod(self.shape)

    # ---- constructors ------------------------------------------------------------------------
    def _filled(self, batch, out):
        return torch.zeros(*batch, *self.shape, **_like(out))

    def zero(self, *shape, out=None):
        return self._filled(shape, out)

    zero_vec = zero

    def rand(self, *shape, out=None, ir=1e-2):
        """Uniform in the cube [-ir, ir]^shape (euclidean.py:50-51)."""
        return self._filled(shape, out).uniform_(-ir, ir)

    def randvec(self, x, norm):
        """Uniform on the sphere of radius `norm` around x (euclidean.py:53-57)."""
        g = torch.randn_like(x)
        return g * (norm / g.norm(dim=self.dims, keepdim=True))

    # ---- the flat geometry ---------------------------------------------------------------------
    def inner(self, x, u, v, keepdim=False):
        return torch.sum(u * v, dim=self.dims, keepdim=keepdim)

    def exp(self, x, u):
        return x + u

    retr = exp

    def log(self, x, y):
        return y - x

    def proju(self, x, u, inplace=False):
        return u

    def egrad2rgrad(self, x, u):
        return u

    def transp(self, x, y, u):
        return u

    def projx(self, x, inplace=False):
        return x
