"""CPU stand-ins for the HIP-backed manifolds, built on the oracle port.  TEST ONLY: they let
the host logic (RiemannianSGD control flow, ManifoldEmbedding, pair-range sharding and its
single all-reduce) run without a GPU.  The product never imports this."""
import torch

from graphembed import _backend as B
from graphembed.manifolds.base import Manifold
from oracle import ref_port as rp


class DoubleManifold(Manifold):

    def __init__(self, port):
        self.port = port

    ndim = property(lambda self: self.port.ndim)
    dim = property(lambda self: 0)

    def zero(self, *shape, out=None):
        raise NotImplementedError

    zero_vec = zero

    def inner(self, x, u, v, keepdim=False):
        return self.port.inner(x, u, v, keepdim)

    def norm(self, x, u, squared=False, keepdim=False):
        return self.port.norm(x, u, squared, keepdim)

    def proju(self, x, u, inplace=False):
        return self.port.proju(x, u)

    def projx(self, x, inplace=False):
        new = self.port.projx(x.detach())
        if inplace:
            x.set_(new)
            return x
        return new

    def egrad2rgrad(self, x, u):
        return self.port.egrad2rgrad(x, u)

    def exp(self, x, u):
        return self.port.exp(x, u)

    def retr(self, x, u):
        return self.port.retr(x, u)

    def log(self, x, y):
        return self.port.log(x, y)

    def transp(self, x, y, u):
        return self.port.transp(x, y, u)

    def dist(self, x, y, squared=False, keepdim=False):
        return self.port.dist(x, y, squared, keepdim)

    def pdist(self, x, squared=False, rows=None):
        full = self.port.pdist(x, squared)
        if rows is None:
            return full
        n = x.shape[0]
        return full[B.pair_offset(n, rows[0]):B.pair_offset(n, rows[1])]

    def rand(self, *shape, out=None, **kw):
        return self.port.rand(shape[0], dtype=torch.float64, **kw)

    def randvec(self, x, norm=1):
        raise NotImplementedError

    def __str__(self):
        return f'cpu double of {type(self.port).__name__}'


def make(name, *args):
    return DoubleManifold(rp.make(name, *args))
