"""Shared machinery of the vector manifolds (Euclidean, Lorentz, Sphere): every
method of the Manifold API routed to the gfx950 kernels of csrc/vec.hip and
csrc/vec_gram.hip.  A point is the flattened trailing `ndim` dimensions."""
import numpy as np
import torch

from graphembed import _backend as B
from graphembed.manifolds.base import Manifold


import os as _os

# MM_VEC_BWD = sym | gram: force the symmetric VALU backward (csrc/vec_sym.hpp) or the matrix-core one (csrc/vec_gram.hip)
_BWD_FORM = _os.environ.get('MM_VEC_BWD', '')


def _pdist_forms(kind, m, n, f32, squared, use_gram):
    """(forward on the matrix cores?, backward on the matrix cores?) for a pdist of n points of dimension m.
    Forward: the Gram kernel serves the inner-product manifolds up to m = 32 (fp32) / 16 (fp64) and n = 32768.
    Backward: up to m = 16 the symmetric VALU kernel (csrc/vec_sym.hpp; Lorentz / sphere: flushed straight into the
    gradient, two launches) beats the matrix-core one — Lorentz(11) n = 4039 forward + backward 44 us against 54 us in
    fp32, 113 against 142 in fp64; the fp32 Euclidean squared distance keeps the matrix cores (67 / 76)."""
    fwd = bool(use_gram and kind in (B.LORENTZ, B.SPHERE) and n <= 32768 and m <= (32 if f32 else 16))
    bwd = n <= 32768 and ((use_gram and kind in (B.LORENTZ, B.SPHERE) and m <= (32 if f32 else 16)) or
                          (f32 and kind == B.EUCLIDEAN and squared and m <= 31))
    if _BWD_FORM == 'sym' or (_BWD_FORM != 'gram' and m <= 16 and (not f32 or kind != B.EUCLIDEAN)):
        bwd = False
    return fwd, bool(bwd)


class _VecPdist(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, kind, m, squared, row_begin, row_end, use_gram):
        B.require_gpu(x)
        lib = B.lib()
        n = x.shape[0]
        xc = x.detach().reshape(n, m).contiguous()
        npairs = B.pair_offset(n, row_end) - B.pair_offset(n, row_begin)
        ctx.save_for_backward(xc)
        ctx.args = (kind, m, squared, row_begin, row_end, x.shape)
        ctx.use_gram = use_gram
        ctx.empty = npairs == 0
        if ctx.empty:
            return xc.new_empty(0)
        with B.on_device(xc.device):
            out = torch.empty(npairs, dtype=xc.dtype, device=xc.device)
            gram, _ = _pdist_forms(kind, m, n, xc.dtype == torch.float32, squared, use_gram)
            name = 'mm_vec_pdist_fwd_gram' if gram else 'mm_vec_pdist_fwd'
            lib.call(name, B.dtype_code(xc), kind, B.ptr(xc), n, m, row_begin, row_end, int(squared),
                     B.ptr(out), B.stream_of(xc))
        return out

    @staticmethod
    def backward(ctx, g):
        xc, = ctx.saved_tensors
        kind, m, squared, row_begin, row_end, shape = ctx.args
        if ctx.empty:
            return (torch.zeros(shape, dtype=xc.dtype, device=xc.device), ) + (None, ) * 6
        lib = B.lib()
        g = g.contiguous()
        n = xc.shape[0]
        dt = B.dtype_code(xc)
        with B.on_device(xc.device):
            grad = torch.empty_like(xc)
            _, mfma = _pdist_forms(kind, m, n, xc.dtype == torch.float32, squared, ctx.use_gram)
            if mfma:
                # matrix-core backward (inner-product manifolds, fp32): W^T X, no workspace
                lib.call('mm_vec_pdist_bwd_gram', dt, kind, B.ptr(xc), B.ptr(g), n, m, row_begin,
                         row_end, int(squared), B.ptr(grad), B.stream_of(xc))
            else:
                ws = torch.empty(lib.raw('mm_vec_pdist_ws_bytes')(dt, n, m), dtype=torch.uint8,
                                 device=xc.device)
                lib.call('mm_vec_pdist_bwd', dt, kind, B.ptr(xc), B.ptr(g), n, m, row_begin, row_end,
                         int(squared), B.ptr(grad), B.ptr(ws), B.stream_of(xc))
        return grad.reshape(shape), None, None, None, None, None, None


class _VecPdistLoss(torch.autograd.Function):
    """loss(target, softplus(scale) * pdist(x)^2) with both gradients from one pass over the
    pairs (mm_vec_pdist_loss) — see graphembed.manifolds.spd._SpdPdistLoss."""

    @staticmethod
    def forward(ctx, x, scale, target, kind, m, spec, row_begin, row_end):
        B.require_gpu(x, target)
        lib = B.lib()
        n = x.shape[0]
        xc = x.detach().reshape(n, m).contiguous()
        dt = B.dtype_code(xc)
        lkind, alpha, eps, terms = spec[:4]
        dyn = spec[4] if len(spec) > 4 else None  # device {alpha, eps} (QuotientLoss.on_device)
        tc = target.detach().to(xc.dtype).contiguous()
        npairs = B.pair_offset(n, row_end) - B.pair_offset(n, row_begin)
        if tc.numel() != npairs:
            raise ValueError(f'target has {tc.numel()} entries, the pair range has {npairs}')
        sc = None if scale is None else scale.detach().to(xc.dtype).reshape(1).contiguous()
        with B.on_device(xc.device):
            ws = torch.empty(lib.raw('mm_vec_pdist_ws_bytes')(dt, n, m), dtype=torch.uint8,
                             device=xc.device)
            out = torch.empty(2, dtype=xc.dtype, device=xc.device)
            grad = torch.empty_like(xc)
            lib.call('mm_vec_pdist_loss', dt, kind, B.LOSS_STRESS if lkind == 'stress' else B.LOSS_QUOTIENT,
                     B.ptr(xc), B.ptr(tc), B.ptr(sc), n, m, row_begin, row_end, alpha, eps, terms, B.dyn_ptr(dyn, xc),
                     B.ptr(out), B.ptr(grad), B.ptr(ws), B.stream_of(xc))
        ctx.grad_x = grad.reshape(x.shape)
        ctx.grad_s = None if scale is None else out[1].reshape(scale.shape).to(scale.dtype)
        return out[0]

    @staticmethod
    def backward(ctx, up):
        gx, gs = B.take_grads(ctx, up, 'grad_x', 'grad_s')
        return gx, gs, None, None, None, None, None, None


class _VecDist(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, y, kind, m, squared):
        B.require_gpu(x, y)
        xc = x.detach().reshape(-1, m).contiguous()
        yc = y.detach().reshape(-1, m).contiguous()
        cnt = xc.shape[0]
        with B.on_device(xc.device):
            out = torch.empty(cnt, dtype=xc.dtype, device=xc.device)
            B.lib().call('mm_vec_dist', B.dtype_code(xc), kind, B.ptr(xc), B.ptr(yc), None, cnt, m,
                         int(squared), B.ptr(out), None, None, B.stream_of(xc))
        ctx.save_for_backward(xc, yc)
        ctx.args = (kind, m, squared, x.shape, y.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        xc, yc = ctx.saved_tensors
        kind, m, squared, xs, ys = ctx.args
        g = g.reshape(-1).contiguous()
        with B.on_device(xc.device):
            gx, gy = torch.empty_like(xc), torch.empty_like(yc)
            B.lib().call('mm_vec_dist', B.dtype_code(xc), kind, B.ptr(xc), B.ptr(yc), B.ptr(g),
                         xc.shape[0], m, int(squared), None, B.ptr(gx), B.ptr(gy), B.stream_of(xc))
        return gx.reshape(xs), gy.reshape(ys), None, None, None


class VectorManifold(Manifold):
    """Base of Euclidean / Lorentz / Sphere.  Subclasses set `_kind` and `shape`."""

    _kind = None
    use_gram = False  # forward pdist through the MFMA Gram kernel (inner-product manifolds)

    @property
    def _m(self):
        # (cached per shape: the product went through numpy on every pdist / dist / map call, ~1.5 us each on the eager path)
        cache = self.__dict__.get('_m_cache')
        if cache is None or cache[0] != self.shape:
            cache = (tuple(self.shape), int(np.prod(self.shape)))
            self.__dict__['_m_cache'] = cache
        return cache[1]

    @property
    def ndim(self):
        return len(self.shape)

    def _batch(self, t):
        return t.shape[:t.ndim - self.ndim]

    # -- per-point maps ------------------------------------------------------
    def _map(self, op, x, u=None, y=None):
        B.require_gpu(x, u, y)
        ts = [t for t in (x, u, y) if t is not None]
        if torch.is_grad_enabled() and any(t.requires_grad for t in ts):
            raise NotImplementedError(
                'exp/log/retr/proj*/transp are optimizer-side maps (torch.no_grad) on the HIP path; '
                'only dist/pdist are differentiable')
        shape = torch.broadcast_shapes(*[t.shape for t in ts])
        m = self._m
        flat = [None if t is None else t.expand(shape).reshape(-1, m).contiguous() for t in (x, u, y)]
        xc = flat[0]
        with B.on_device(xc.device):
            out = torch.empty_like(xc)
            B.lib().call('mm_vec_map', B.dtype_code(xc), self._kind, op, B.ptr(flat[0]), B.ptr(flat[1]),
                         B.ptr(flat[2]), xc.shape[0], m, B.ptr(out), B.stream_of(xc))
        return out.reshape(shape)

    def _set_or_return(self, t, new, inplace):
        if not inplace:
            return new
        t.set_(new)
        return t

    def norm(self, x, u, squared=False, keepdim=False):  # base.py:29-33
        if torch.is_grad_enabled() and u.requires_grad:
            return super().norm(x, u, squared, keepdim)
        B.require_gpu(u)
        m = self._m
        uc = u.detach().reshape(-1, m).contiguous()
        with B.on_device(uc.device):
            out = torch.empty(uc.shape[0], dtype=uc.dtype, device=uc.device)
            B.lib().call('mm_vec_norm', B.dtype_code(uc), self._kind, B.ptr(uc), uc.shape[0], m,
                         int(squared), B.ptr(out), B.stream_of(uc))
        out = out.reshape(self._batch(u))
        return out.reshape(*out.shape, *((1, ) * self.ndim)) if keepdim else out

    def proju(self, x, u, inplace=False):
        return self._set_or_return(u, self._map(B.VEC_PROJU, x, u), inplace)

    def projx(self, x, inplace=False):
        return self._set_or_return(x, self._map(B.VEC_PROJX, x.detach() if inplace else x), inplace)

    def egrad2rgrad(self, x, u, inplace=False):
        return self._set_or_return(u, self._map(B.VEC_EGRAD2RGRAD, x, u), inplace)

    def exp(self, x, u):
        return self._map(B.VEC_EXP, x, u)

    def retr(self, x, u):
        return self._map(B.VEC_RETR, x, u)

    def log(self, x, y):
        return self._map(B.VEC_LOG, x, y)

    def transp(self, x, y, u):
        return self._map(B.VEC_TRANSP, x, u, y)

    def rsgd_step(self, x, egrad, *, lr, max_grad_norm=None, exact=False, inplace=False):
        """Fused momentum-free RiemannianSGD update (optim/rsgd.py:63-68,82); `inplace=True` writes
        the new points over `x` (each thread reads its whole point before writing it)."""
        B.require_gpu(x, egrad)
        m = self._m
        xd = x.detach()
        inplace = inplace and xd.is_contiguous()
        xc = xd.reshape(-1, m).contiguous()
        gc = egrad.detach().reshape(-1, m).contiguous()
        with B.on_device(xc.device):
            out = xc if inplace else torch.empty_like(xc)
            B.lib().call('mm_vec_rsgd_step', B.dtype_code(xc), self._kind, B.ptr(xc), B.ptr(gc),
                         xc.shape[0], m, float(lr),
                         -1.0 if max_grad_norm is None else float(max_grad_norm), int(bool(exact)),
                         B.ptr(out), B.stream_of(xc))
        return x if inplace else out.reshape(x.shape)

    def rsgd_momentum_step(self, x, egrad, buf, *, lr, momentum, dampening, max_grad_norm=None, exact=False,
                           inplace=False):
        """Fused heavy-ball RiemannianSGD update (optim/rsgd.py:70-80): `buf` (the momentum buffer) is
        updated and transported in place; returns the new points, or None when not eligible."""
        return _vec_momentum(self._kind, self._m, x, egrad, buf, lr, momentum, dampening, max_grad_norm, exact,
                             inplace)

    def radam_step(self, x, egrad, exp_avg, exp_avg_sq, step, ticket, *, lr, betas, nc, eps, max_grad_norm=None,
                   exact=False, inplace=False):
        """Fused RiemannianAdam update (optim/radam.py:62-98) in one launch: moments updated in place,
        `step` (device fp64 scalar) advanced by the kernel; returns the new points (`x` itself when
        `inplace`), or None when the tensors are not eligible."""
        return _vec_radam(self._kind, self._m, x, egrad, exp_avg, exp_avg_sq, step, ticket, lr, betas, nc, eps,
                          max_grad_norm, exact, inplace)

    # -- distances -------------------------------------------------------------
    def dist(self, x, y, squared=False, keepdim=False):
        shape = torch.broadcast_shapes(x.shape, y.shape)
        d = _VecDist.apply(x.expand(shape), y.expand(shape), self._kind, self._m, squared)
        d = d.reshape(shape[:len(shape) - self.ndim])
        return d.reshape(*d.shape, *((1, ) * self.ndim)) if keepdim else d

    def pdist(self, x, squared=False, rows=None):
        """All-pairs distances, row-major upper triangle (base.py:59-63).  `rows` selects the
        pair-list slice of one shard (graphembed.parallel)."""
        assert x.ndim == self.ndim + 1
        rb, re = (0, x.shape[0]) if rows is None else rows
        ext = B.autograd_ext()
        if ext is not None:   # the same C-ABI calls as _VecPdist, issued by C++ autograd nodes (csrc_torch/mm_autograd.cpp)
            fwd_gram, bwd_gram = _pdist_forms(self._kind, self._m, x.shape[0], x.dtype == torch.float32, bool(squared),
                                              self.use_gram)
            try:
                return ext.vec_pdist(x, self._kind, self._m, bool(squared), int(rb), int(re), fwd_gram, bwd_gram)
            except RuntimeError as e:
                if 'runs on MI355X only' in str(e) or 'failed:' in str(e):
                    raise B.BackendError(str(e)) from None
                raise
        return _VecPdist.apply(x, self._kind, self._m, squared, rb, re, self.use_gram)

    def pdist_loss(self, x, scale, target, spec, rows=None):
        """Fused `objective(target, softplus(scale) * pdist(x, squared=True))` with its gradients in
        one pass; `spec` comes from `objective_fn.fused_spec(epoch=, alpha=)`."""
        assert x.ndim == self.ndim + 1
        rb, re = (0, x.shape[0]) if rows is None else rows
        return _VecPdistLoss.apply(x, scale, target, self._kind, self._m, spec, rb, re)


def _vec_radam(kind, m, x, egrad, exp_avg, exp_avg_sq, step, ticket, lr, betas, nc, eps, max_grad_norm, exact,
               inplace):
    ok = (x.is_cuda and x.dtype in (torch.float32, torch.float64) and 1 <= m <= B.lib().raw('mm_vec_max_dim')() and x.numel() > 0
          and exp_avg.is_contiguous() and exp_avg_sq.is_contiguous() and exp_avg.dtype == x.dtype
          and exp_avg_sq.dtype == x.dtype and exp_avg.shape == x.shape and exp_avg_sq.shape == x.shape)
    if not ok:
        return None
    xd = x.detach()
    inplace = inplace and xd.is_contiguous()
    xc = xd.reshape(-1, m).contiguous()
    gc = egrad.detach().reshape(-1, m).to(xc.dtype).contiguous()
    with B.on_device(xc.device):
        out = xc if inplace else torch.empty_like(xc)
        B.lib().call('mm_vec_radam_step', B.dtype_code(xc), kind, B.ptr(xc), B.ptr(gc), B.ptr(exp_avg),
                     B.ptr(exp_avg_sq), B.ptr(step), B.ptr(ticket), xc.shape[0], m, float(lr), float(betas[0]),
                     float(betas[1] if betas[1] is not None else 0.0), int(bool(nc)), float(eps),
                     -1.0 if max_grad_norm is None else float(max_grad_norm), int(bool(exact)), B.ptr(out),
                     B.stream_of(xc))
    return x if inplace else out.reshape(x.shape)


def _vec_momentum(kind, m, x, egrad, buf, lr, momentum, dampening, max_grad_norm, exact, inplace):
    ok = (x.is_cuda and x.dtype in (torch.float32, torch.float64) and 1 <= m <= B.lib().raw('mm_vec_max_dim')() and x.numel() > 0
          and buf.is_contiguous() and buf.dtype == x.dtype and buf.shape == x.shape)
    if not ok:
        return None
    xd = x.detach()
    inplace = inplace and xd.is_contiguous()
    xc = xd.reshape(-1, m).contiguous()
    gc = egrad.detach().reshape(-1, m).to(xc.dtype).contiguous()
    with B.on_device(xc.device):
        out = xc if inplace else torch.empty_like(xc)
        B.lib().call('mm_vec_rsgd_momentum_step', B.dtype_code(xc), kind, B.ptr(xc), B.ptr(gc), B.ptr(buf),
                     xc.shape[0], m, float(lr), float(momentum), float(dampening),
                     -1.0 if max_grad_norm is None else float(max_grad_norm), int(bool(exact)), B.ptr(out),
                     B.stream_of(xc))
    return x if inplace else out.reshape(x.shape)
