"""The `Manifold` plugin API — the drop-in boundary of this package on the Python side.

Method names, argument meaning and defaults are those of the reference's abstract class
(graphembed/graphembed/manifolds/base.py:7-81): a caller written against the reference
(`ManifoldEmbedding`, the optimizers, `monitor`, the YAML factories) runs unchanged.  Conventions:

* a *point* occupies the trailing `ndim` dimensions of a tensor, everything in front is batch;
* `out=` only donates dtype/device (that is all the reference uses it for);
* value clamps are gradient-transparent (`t.data.clamp_` in the reference): `value_clamp`;
* on the HIP path `dist` / `pdist` are differentiable, the optimizer-side maps
  (`exp`, `retr`, `log`, `proj*`, `transp`, `egrad2rgrad`) run under `torch.no_grad()`.

Extensions a concrete manifold MAY offer (callers probe with `getattr`):
  `pdist(x, squared, rows=(r0, r1))`  the pair-list slice of one shard (graphembed.parallel)
  `pdist_loss(x, scale, target, spec, rows)`  loss and gradients in one pass (fused objective)
  `rsgd_step(x, egrad, lr=, max_grad_norm=, exact=)`  fused momentum-free RSGD update
"""
import abc

import torch

from graphembed.utils import EPS


def value_clamp(t, lo=None, hi=None):
    """Clamp the VALUE of `t`; the gradient passes through unchanged."""
    return _PassThroughClamp.apply(t, lo, hi)


def _like(out):
    """dtype / device carried by an `out=` argument."""
    return {} if out is None else {'dtype': out.dtype, 'device': out.device}


def pair_index(n, device=None, rows=None):
    """(i, j) of the row-major upper triangle — the order of every pair vector in this package —
    optionally restricted to the rows [r0, r1) of one shard."""
    ij = torch.triu_indices(n, n, 1, device=device)
    if rows is None:
        return ij[0], ij[1]
    keep = (ij[0] >= rows[0]) & (ij[0] < rows[1])
    return ij[0][keep], ij[1][keep]


class _PassThroughClamp(torch.autograd.Function):

    @staticmethod
    def forward(ctx, t, lo, hi):
        return t.clamp(min=lo, max=hi)

    @staticmethod
    def backward(ctx, g):
        return g, None, None


class Manifold(abc.ABC):
    # ---- what a manifold must define ------------------------------------------------------------
    @property
    @abc.abstractmethod
    def ndim(self):
        """Number of trailing tensor dimensions that make up one point."""

    @property
    @abc.abstractmethod
    def dim(self):
        """Intrinsic dimension of the manifold."""

    @abc.abstractmethod
    def zero(self, *shape, out=None):
        """The origin, batch shape `shape`."""

    @abc.abstractmethod
    def zero_vec(self, *shape, out=None):
        """Zero tangent vectors, batch shape `shape`."""

    @abc.abstractmethod
    def inner(self, x, u, v, keepdim=False):
        """Riemannian inner product <u, v>_x."""

    @abc.abstractmethod
    def proju(self, x, u, inplace=False):
        """Ambient vector -> tangent space at x."""

    @abc.abstractmethod
    def projx(self, x, inplace=False):
        """Ambient point -> manifold."""

    @abc.abstractmethod
    def exp(self, x, u):
        """Exponential map."""

    @abc.abstractmethod
    def log(self, x, y):
        """Logarithmic map."""

    @abc.abstractmethod
    def rand(self, *shape, out=None):
        """Random points (the embedding's initialisation)."""

    @abc.abstractmethod
    def randvec(self, x, norm=1):
        """Random tangent vectors of the given norm."""

    @abc.abstractmethod
    def __str__(self):
        ...

    # ---- defaults in terms of the above (base.py:29-66 of the reference) --------------------------
    def rand_uniform(self, *shape, out=None):
        raise NotImplementedError(f'{type(self).__name__} has no uniform sampler')

    def norm(self, x, u, squared=False, keepdim=False):
        sq = value_clamp(self.inner(x, u, u, keepdim), EPS[u.dtype])
        return sq if squared else sq.sqrt()

    def egrad2rgrad(self, x, u):
        return self.proju(x, u)

    def retr(self, x, u):
        return self.exp(x, u)

    def transp(self, x, y, u):
        return self.proju(y, u)

    def dist(self, x, y, squared=False, keepdim=False):
        return self.norm(x, self.log(x, y), squared, keepdim)

    def pdist(self, x, squared=False, rows=None):
        """All pairs of the batch `x`, row-major upper triangle; generic form through `dist`
        (the concrete manifolds of this package override it with pair kernels)."""
        if x.ndim != self.ndim + 1:
            raise ValueError(f'pdist expects a batch of points, got a tensor of shape {tuple(x.shape)}')
        i, j = pair_index(x.shape[0], x.device, rows)
        return self.dist(x[i], x[j], squared=squared, keepdim=False)
