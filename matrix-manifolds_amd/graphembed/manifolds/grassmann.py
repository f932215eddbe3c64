"""Grassmann Gr(n,p) and Stiefel St(n,p) on the gfx950 kernels of csrc/mat.hip — counterparts
of graphembed/graphembed/manifolds/grassmann.py:10-116 and stiefel.py:7-93 (the reference ships
every QR/SVD of these maps to the CPU, linalg/torch_batch.py:94-121)."""
import torch

from graphembed import _backend as B
from graphembed.manifolds.base import Manifold, _like


class _GrassPdist(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, N, p, squared, row_begin, row_end):
        B.require_gpu(x)
        xc = x.detach().contiguous()
        n = xc.shape[0]
        npairs = B.pair_offset(n, row_end) - B.pair_offset(n, row_begin)
        ctx.save_for_backward(xc)
        ctx.args = (N, p, squared, row_begin, row_end)
        ctx.empty = npairs == 0
        if ctx.empty:
            return xc.new_empty(0)
        with B.on_device(xc.device):
            out = torch.empty(npairs, dtype=xc.dtype, device=xc.device)
            B.lib().call('mm_grass_pdist_fwd', B.dtype_code(xc), B.ptr(xc), n, N, p, row_begin, row_end,
                         int(squared), B.ptr(out), B.stream_of(xc))
        return out

    @staticmethod
    def backward(ctx, g):
        xc, = ctx.saved_tensors
        N, p, squared, row_begin, row_end = ctx.args
        if ctx.empty:
            return (torch.zeros_like(xc), ) + (None, ) * 5
        lib = B.lib()
        g = g.contiguous()
        n = xc.shape[0]
        dt = B.dtype_code(xc)
        with B.on_device(xc.device):
            ws = torch.empty(lib.raw('mm_grass_pdist_ws_bytes')(dt, n, N, p), dtype=torch.uint8, device=xc.device)
            grad = torch.empty_like(xc)
            lib.call('mm_grass_pdist_bwd', dt, B.ptr(xc), B.ptr(g), n, N, p, row_begin, row_end, int(squared),
                     B.ptr(grad), B.ptr(ws), B.stream_of(xc))
        return grad, None, None, None, None, None


class _GrassDist(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, y, N, p, squared):
        B.require_gpu(x, y)
        xc = x.detach().reshape(-1, N, p).contiguous()
        yc = y.detach().reshape(-1, N, p).contiguous()
        with B.on_device(xc.device):
            out = torch.empty(xc.shape[0], dtype=xc.dtype, device=xc.device)
            B.lib().call('mm_grass_dist', B.dtype_code(xc), B.ptr(xc), B.ptr(yc), None, xc.shape[0], N, p,
                         int(squared), B.ptr(out), None, None, B.stream_of(xc))
        ctx.save_for_backward(xc, yc)
        ctx.args = (N, p, squared, x.shape, y.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        xc, yc = ctx.saved_tensors
        N, p, squared, xs, ys = ctx.args
        g = g.reshape(-1).contiguous()
        with B.on_device(xc.device):
            gx, gy = torch.empty_like(xc), torch.empty_like(yc)
            B.lib().call('mm_grass_dist', B.dtype_code(xc), B.ptr(xc), B.ptr(yc), B.ptr(g), xc.shape[0], N, p,
                         int(squared), None, B.ptr(gx), B.ptr(gy), B.stream_of(xc))
        return gx.reshape(xs), gy.reshape(ys), None, None, None


class _MatrixManifold(Manifold):
    _kind = None

    def __init__(self, n, p, retr='svd'):
        self.n = n
        self.p = p
        if retr == 'qr':
            self.retr = self.retr_qr_
        elif retr == 'svd':
            self.retr = self.retr_svd_
        else:
            raise ValueError('Unknown retraction type {}'.format(retr))

    @property
    def ndim(self):
        return 2

    def zero(self, *shape, out=None):
        return torch.eye(self.n, self.p, **_like(out)).repeat(*shape, 1, 1)

    def zero_vec(self, *shape, out=None):
        return torch.zeros(*shape, self.n, self.p, **_like(out))

    def inner(self, x, u, v, keepdim=False):
        return (u * v).sum((-2, -1), keepdim=keepdim)

    def _map(self, op, x, u=None):
        B.require_gpu(x, u)
        ts = [t for t in (x, u) if t is not None]
        if torch.is_grad_enabled() and any(t.requires_grad for t in ts):
            raise NotImplementedError('projections / retractions / exp / log run under torch.no_grad on the '
                                      'HIP path; only dist/pdist are differentiable')
        shape = torch.broadcast_shapes(*[t.shape for t in ts])
        xc = x.expand(shape).reshape(-1, self.n, self.p).contiguous()
        uc = None if u is None else u.expand(shape).reshape(-1, self.n, self.p).contiguous()
        with B.on_device(xc.device):
            out = torch.empty_like(xc)
            B.lib().call('mm_mat_map', B.dtype_code(xc), self._kind, op, B.ptr(xc), B.ptr(uc), xc.shape[0],
                         self.n, self.p, B.ptr(out), B.stream_of(xc))
        return out.reshape(shape)

    def proju(self, x, u, inplace=False):
        new = self._map(B.MAT_PROJU, x, u)
        if not inplace:
            return new
        u.set_(new)
        return u

    def retr_qr_(self, x, u):
        return self._map(B.MAT_RETR_QR, x, u)

    def retr_svd_(self, x, u):
        return self._map(B.MAT_RETR_SVD, x, u)

    def rand(self, *shape, out=None, ir=1e-2):
        x = self.zero(*shape, out=out)
        with torch.no_grad():
            return self._rand_step(x, self.randvec(x, norm=ir))

    def rand_uniform(self, *shape, out=None):
        with torch.no_grad():
            return self.projx(torch.randn(*shape, self.n, self.p, **_like(out)), inplace=True)

    def randvec(self, x, norm):
        with torch.no_grad():
            u = self.proju(x, torch.randn_like(x))
        return u.div_(u.norm(dim=(-2, -1), keepdim=True)).mul_(norm)


class Grassmann(_MatrixManifold):
    _kind = B.GRASSMANN

    def __init__(self, n, p, retr='svd', requires_grad=True):
        super().__init__(n, p, retr)
        self.requires_grad = requires_grad

    @property
    def dim(self):
        return self.p * (self.n - self.p)

    def projx(self, x, inplace=False):  # grassmann.py:55-61
        new = self._map(B.MAT_PROJX, x.detach() if inplace else x)
        if not inplace:
            return new
        x.set_(new)
        return x

    def exp(self, x, u):  # grassmann.py:63-69
        return self._map(B.MAT_EXP, x, u)

    def log(self, x, y):  # grassmann.py:82-89
        return self._map(B.MAT_LOG, x, y)

    def _rand_step(self, x, u):
        return self.exp(x, u)

    def dist(self, x, y, squared=False, keepdim=False):  # grassmann.py:91-96
        shape = torch.broadcast_shapes(x.shape, y.shape)
        d = _GrassDist.apply(x.expand(shape), y.expand(shape), self.n, self.p, squared).reshape(shape[:-2])
        return d.reshape(*d.shape, 1, 1) if keepdim else d

    def pdist(self, x, squared=False, rows=None):  # base.py:59-63
        assert x.ndim == 3
        rb, re = (0, x.shape[0]) if rows is None else rows
        return _GrassPdist.apply(x, self.n, self.p, squared, rb, re)

    def __str__(self):
        return 'Grassmann manifold of {}x{} matrices'.format(self.n, self.p)


class Stiefel(_MatrixManifold):
    _kind = B.STIEFEL

    @property
    def dim(self):
        return self.p * self.n - self.p * (self.p + 1) // 2

    def _orthonormalize(self, x):  # stiefel.py:47-50: Q of QR with columns signed by diag(R)
        return self._map(B.MAT_PROJX, x)

    def projx(self, x, inplace=False):  # stiefel.py:53-57 (returns x itself when not inplace, as shipped)
        if inplace:
            x.set_(self._orthonormalize(x.detach()))
        return x

    def exp(self, x, u):  # stiefel.py:59-60 (sic: returns the exception class)
        return NotImplementedError

    def log(self, x, y):
        return NotImplementedError

    def dist(self, x, y, squared=False, keepdim=False):
        return NotImplementedError

    def _rand_step(self, x, u):
        return self.retr(x, u)

    def rand_uniform(self, *shape, out=None):
        with torch.no_grad():
            return self._orthonormalize(torch.randn(*shape, self.n, self.p, **_like(out)))

    def __str__(self):
        return 'Stiefel manifold of {}x{} matrices'.format(self.n, self.p)
