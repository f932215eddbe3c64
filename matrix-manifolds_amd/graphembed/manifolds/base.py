"""The `Manifold` plugin API — the drop-in boundary of this package.

Same method names, argument meaning and defaults as the reference's abstract
class (graphembed/graphembed/manifolds/base.py:7-81).  Concrete manifolds route
their arithmetic to the gfx950 library through `graphembed._backend`.
"""
import abc

import torch

from graphembed.utils import EPS


class _ValueClamp(torch.autograd.Function):
    """The reference's `t.data.clamp_(lo, hi)` idiom: clamp the value, pass the gradient."""

    @staticmethod
    def forward(ctx, t, lo, hi):
        return t.clamp(min=lo, max=hi)

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def value_clamp(t, lo=None, hi=None):
    return _ValueClamp.apply(t, lo, hi)


class Manifold(metaclass=abc.ABCMeta):

    @property
    @abc.abstractmethod
    def ndim(self):
        """Number of trailing dimensions that make up one point."""

    @property
    @abc.abstractmethod
    def dim(self):
        """Intrinsic dimension."""

    @abc.abstractmethod
    def zero(self, *shape, out=None):
        pass

    @abc.abstractmethod
    def zero_vec(self, *shape, out=None):
        pass

    @abc.abstractmethod
    def inner(self, x, u, v, keepdim=False):
        pass

    def norm(self, x, u, squared=False, keepdim=False):  # base.py:29-33
        nsq = value_clamp(self.inner(x, u, u, keepdim), EPS[u.dtype])
        return nsq if squared else nsq.sqrt()

    @abc.abstractmethod
    def proju(self, x, u, inplace=False):
        pass

    @abc.abstractmethod
    def projx(self, x, inplace=False):
        pass

    def egrad2rgrad(self, x, u):  # base.py:43-44
        return self.proju(x, u)

    @abc.abstractmethod
    def exp(self, x, u):
        pass

    def retr(self, x, u):  # base.py:49-50
        return self.exp(x, u)

    @abc.abstractmethod
    def log(self, x, y):
        pass

    def dist(self, x, y, squared=False, keepdim=False):  # base.py:56-57
        return self.norm(x, self.log(x, y), squared, keepdim)

    def pdist(self, x, squared=False):  # base.py:59-63
        assert x.ndim == self.ndim + 1
        n = x.shape[0]
        m = torch.triu_indices(n, n, 1, device=x.device)
        return self.dist(x[m[0]], x[m[1]], squared=squared, keepdim=False)

    def transp(self, x, y, u):  # base.py:65-66
        return self.proju(y, u)

    @abc.abstractmethod
    def rand(self, *shape, out=None):
        pass

    def rand_uniform(self, *shape, out=None):
        raise NotImplementedError

    @abc.abstractmethod
    def randvec(self, x, norm=1):
        pass

    @abc.abstractmethod
    def __str__(self):
        pass


def _like(out):
    """dtype/device carried by an `out=` tensor (the reference only uses `out` for that)."""
    if out is None:
        return {}
    return dict(dtype=out.dtype, device=out.device)
