"""Hyperboloid model H^{n-1} — counterpart of graphembed/graphembed/manifolds/lorentz.py:9-98."""
import torch

from graphembed import _backend as B
from graphembed.manifolds.base import _like
from graphembed.manifolds.sphere import Sphere
from graphembed.manifolds.vector import VectorManifold


def ldot(u, v, keepdim=False):
    """Minkowski inner product -u0 v0 + sum_k u_k v_k (lorentz.py:101-122)."""
    uv = u * v
    return uv[..., 1:].sum(-1, keepdim=keepdim) - uv[..., :1].sum(-1, keepdim=keepdim)


class Lorentz(VectorManifold):
    _kind = B.LORENTZ
    use_gram = True

    def __init__(self, n):
        self.n = n
        self.shape = (n, )
        self.sphere = Sphere(self.n - 1)

    @staticmethod
    def to_poincare_ball(x):
        d = x.shape[-1] - 1
        return x.narrow(-1, 1, d) / (x.narrow(-1, 0, 1) + 1)

    @property
    def dim(self):
        return self.n - 1

    def zero(self, *shape, out=None):
        x = torch.zeros(*shape, self.n, **_like(out))
        x[..., 0] = 1
        return x

    def zero_vec(self, *shape, out=None):
        return torch.zeros(*shape, self.n, **_like(out))

    def inner(self, x, u, v, keepdim=False):
        return ldot(u, v, keepdim=keepdim)

    def rand(self, *shape, out=None, ir=1e-2):  # lorentz.py:84-86
        x = torch.empty(*shape, self.n, **_like(out)).uniform_(-ir, ir)
        with torch.no_grad():
            return self.projx(x)

    def randvec(self, x, norm=1):  # lorentz.py:88-95
        shape = x.shape[:-1]
        dirs = self.sphere.rand_uniform(*shape, out=x)
        vs = torch.cat([torch.zeros(*shape, 1, dtype=x.dtype, device=x.device), dirs], dim=-1).mul_(norm)
        with torch.no_grad():
            return self.transp(self.zero(*shape, out=x), x, vs)

    def __str__(self):
        return 'Lorentzian space of dimension {}'.format(self.n)
