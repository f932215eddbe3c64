"""SPD(n) with the affine-invariant metric, on the gfx950 kernels.

Mirror of graphembed/graphembed/manifolds/spd.py:21-243 (same constructor,
methods, shapes and clamps).  Differences, all documented in DESIGN.md §6:

* one eigensolver for every n (cyclic Jacobi in registers) instead of the
  eps-fudged closed forms for n=2,3 and a CPU LAPACK offload for n>=4 —
  `fast_symeig` / `fast_chol` are accepted and ignored;
* the gradient of `pdist`/`dist` is the symmetric part of what the reference's
  autograd returns (the reference reads only one triangle and yields a
  non-symmetric matrix; everything downstream symmetrises it, spd.py:119-135);
* `use_stein_div=True` rebinds `dist` / `pdist` to the Stein divergence (spd.py:51-53, 183-194, 246-295),
  also on pair kernels (csrc/spd_stein.hpp); the reference's dense (n, n, d, d) backward is gone.
"""
import math

import torch

from graphembed import _backend as B
from graphembed.manifolds.base import Manifold, _like
from graphembed.utils import _to_square, nnp1d2_to_n, squareform0


def _flat(t, n):
    """(…, n, n) -> contiguous (m, n, n)."""
    return t.reshape(-1, n, n).contiguous()


class _SpdPdist(torch.autograd.Function):
    """pdist over rows [row_begin,row_end) of the pair list; backward gives the
    full-shape partial gradient of this shard."""

    @staticmethod
    def forward(ctx, x, n_mat, squared, wmin, wmax, row_begin, row_end, check_pd):
        B.require_gpu(x)
        lib = B.lib()
        xc = x.detach().contiguous()
        n = xc.shape[0]
        dt = B.dtype_code(xc)
        npairs = B.pair_offset(n, row_end) - B.pair_offset(n, row_begin)
        ctx.empty = npairs == 0
        if ctx.empty:
            ctx.save_for_backward(xc)
            return xc.new_empty(0)
        with B.on_device(xc.device):
            ws = torch.empty(lib.raw('mm_spd_pdist_ws_bytes')(dt, n, n_mat), dtype=torch.uint8,
                             device=xc.device)
            out = torch.empty(npairs, dtype=xc.dtype, device=xc.device)
            lib.call('mm_spd_pdist_fwd', dt, B.ptr(xc), n, n_mat, row_begin, row_end, int(squared),
                     wmin, wmax, B.ptr(out), B.ptr(ws), 0, B.stream_of(xc))
            if check_pd:
                import ctypes
                st = ctypes.c_int(0)
                lib.call('mm_spd_status', B.ptr(ws), n, ctypes.byref(st), B.stream_of(xc))
                if st.value:
                    raise torch.linalg.LinAlgError(
                        f'pdist: {st.value} input matrices are not positive-definite')
        ctx.save_for_backward(xc)
        ctx.ws = ws
        ctx.args = (n_mat, squared, wmin, wmax, row_begin, row_end)
        return out

    @staticmethod
    def backward(ctx, g):
        xc, = ctx.saved_tensors
        if ctx.empty:
            return (torch.zeros_like(xc), ) + (None, ) * 7
        n_mat, squared, wmin, wmax, row_begin, row_end = ctx.args
        lib = B.lib()
        g = g.contiguous()
        n = xc.shape[0]
        with B.on_device(xc.device):
            grad = torch.empty_like(xc)
            lib.call('mm_spd_pdist_bwd', B.dtype_code(xc), B.ptr(xc), B.ptr(g), n, n_mat, row_begin,
                     row_end, int(squared), wmin, wmax, B.ptr(grad), B.ptr(ctx.ws),
                     B.MM_WS_PREPARED, B.stream_of(xc))
        return grad, None, None, None, None, None, None, None


_GATHER_BYTES = 256 << 20   # budget of one gathered chunk (both sides) of the narrow-window pdist below


def _pair_row_chunks(n, rb, re, max_pairs, device):
    """The pairs of rows [rb, re) in pair-vector order, cut into chunks of whole rows of at most `max_pairs` pairs (a single
    longer row is a chunk of its own): yields (i, j, lo, hi) — node index vectors of the chunk and its slice [lo, hi) of
    the range's pair vector.  Only the chunk's indices exist at any time (not triu_indices(n, n))."""
    r0, base = rb, B.pair_offset(n, rb)
    re = min(re, n - 1)
    while r0 < re:
        r1 = r0 + 1
        while r1 < re and B.pair_offset(n, r1 + 1) - B.pair_offset(n, r0) <= max_pairs:
            r1 += 1
        rows = torch.arange(r0, r1, device=device)
        counts = (n - 1) - rows
        i = torch.repeat_interleave(rows, counts)
        first = torch.cumsum(counts, 0) - counts                  # offset of each row's first pair inside the chunk
        j = torch.arange(i.numel(), device=device) - first[i - r0] + i + 1
        lo = B.pair_offset(n, r0) - base
        yield i, j, lo, lo + i.numel()
        r0 = r1


class _SpdPdistGathered(torch.autograd.Function):
    """`pdist` under an eigenvalue window narrower than [1e-6, 1e6] for SPD(n >= 3): the element-wise kernels
    (mm_spd_dist_fwd / _bwd, which honour any window) over the gathered pairs — the reference's own Manifold.pdist
    (base.py:59-63) — in CHUNKS of whole rows of the requested range: one chunk's index vectors and gathered operands exist
    at a time (256 MB), where `x[iu[0]]`, `x[iu[1]]` of all pairs plus triu_indices(n, n) were ~1 GB at n = 5000, d = 3 even
    for a small row shard, and the backward scatter-adds per chunk instead of through an index_put over all pairs."""

    @staticmethod
    def forward(ctx, x, n_mat, squared, wmin, wmax, rb, re):
        B.require_gpu(x)
        xc = x.detach().contiguous()
        n = xc.shape[0]
        npairs = B.pair_offset(n, re) - B.pair_offset(n, rb)
        max_pairs = max(1, _GATHER_BYTES // (2 * n_mat * n_mat * xc.element_size()))
        with B.on_device(xc.device):
            out = torch.empty(npairs, dtype=xc.dtype, device=xc.device)
            for i, j, lo, hi in _pair_row_chunks(n, rb, re, max_pairs, xc.device):
                xi, xj = xc.index_select(0, i), xc.index_select(0, j)
                B.lib().call('mm_spd_dist_fwd', B.dtype_code(xc), B.ptr(xi), B.ptr(xj), hi - lo, n_mat, int(squared), wmin, wmax,
                             B.ptr(out[lo:hi]), B.stream_of(xc))
        ctx.save_for_backward(xc)
        ctx.args = (n_mat, squared, wmin, wmax, rb, re, max_pairs)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        xc, = ctx.saved_tensors
        n_mat, squared, wmin, wmax, rb, re, max_pairs = ctx.args
        g = g.contiguous()
        n = xc.shape[0]
        with B.on_device(xc.device):
            grad = torch.zeros_like(xc)
            for i, j, lo, hi in _pair_row_chunks(n, rb, re, max_pairs, xc.device):
                xi, xj = xc.index_select(0, i), xc.index_select(0, j)
                gi, gj = torch.empty_like(xi), torch.empty_like(xj)
                B.lib().call('mm_spd_dist_bwd', B.dtype_code(xc), B.ptr(xi), B.ptr(xj), B.ptr(g[lo:hi]), hi - lo, n_mat,
                             int(squared), wmin, wmax, B.ptr(gi), B.ptr(gj), B.stream_of(xc))
                grad.index_add_(0, i, gi)
                grad.index_add_(0, j, gj)
        return grad, None, None, None, None, None, None


class _SpdPdistLoss(torch.autograd.Function):
    """loss(target, softplus(scale) * pdist(x)^2) and both gradients from ONE pass over the
    pairs (mm_spd_pdist_loss): what train.py:213-217 + modules.py:84-88 + objectives.py:16-45
    compute in the reference, without materialising the pair vector of distances."""

    @staticmethod
    def forward(ctx, x, scale, target, n_mat, spec, wmin, wmax, row_begin, row_end):
        B.require_gpu(x, target)
        lib = B.lib()
        xc = x.detach().contiguous()
        n = xc.shape[0]
        dt = B.dtype_code(xc)
        kind, alpha, eps, terms = spec[:4]
        dyn = spec[4] if len(spec) > 4 else None  # device {alpha, eps} (QuotientLoss.on_device)
        tc = target.detach().to(xc.dtype).contiguous()
        npairs = B.pair_offset(n, row_end) - B.pair_offset(n, row_begin)
        if tc.numel() != npairs:
            raise ValueError(f'target has {tc.numel()} entries, the pair range has {npairs}')
        sc = None if scale is None else scale.detach().to(xc.dtype).reshape(1).contiguous()
        with B.on_device(xc.device):
            ws = torch.empty(lib.raw('mm_spd_pdist_ws_bytes')(dt, n, n_mat), dtype=torch.uint8,
                             device=xc.device)
            out = torch.empty(2, dtype=xc.dtype, device=xc.device)
            grad = torch.empty_like(xc)
            lib.call('mm_spd_pdist_loss', dt, B.LOSS_STRESS if kind == 'stress' else B.LOSS_QUOTIENT,
                     B.ptr(xc), B.ptr(tc), B.ptr(sc), n, n_mat, row_begin, row_end, alpha, eps, terms, B.dyn_ptr(dyn, xc),
                     wmin, wmax, B.ptr(out), B.ptr(grad), B.ptr(ws), 0, B.stream_of(xc))
        ctx.grad_x = grad
        ctx.grad_s = None if scale is None else out[1].reshape(scale.shape).to(scale.dtype)
        return out[0]

    @staticmethod
    def backward(ctx, up):
        gx, gs = B.take_grads(ctx, up, 'grad_x', 'grad_s')
        return gx, gs, None, None, None, None, None, None, None


class _SpdDist(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, y, n_mat, squared, wmin, wmax):
        B.require_gpu(x, y)
        xc, yc = _flat(x.detach(), n_mat), _flat(y.detach(), n_mat)
        m = xc.shape[0]
        with B.on_device(xc.device):
            out = torch.empty(m, dtype=xc.dtype, device=xc.device)
            B.lib().call('mm_spd_dist_fwd', B.dtype_code(xc), B.ptr(xc), B.ptr(yc), m, n_mat,
                         int(squared), wmin, wmax, B.ptr(out), B.stream_of(xc))
        ctx.save_for_backward(xc, yc)
        ctx.args = (n_mat, squared, wmin, wmax, x.shape, y.shape)
        return out.reshape(x.shape[:-2])

    @staticmethod
    def backward(ctx, g):
        xc, yc = ctx.saved_tensors
        n_mat, squared, wmin, wmax, xs, ys = ctx.args
        g = g.reshape(-1).contiguous()
        with B.on_device(xc.device):
            gx, gy = torch.empty_like(xc), torch.empty_like(yc)
            B.lib().call('mm_spd_dist_bwd', B.dtype_code(xc), B.ptr(xc), B.ptr(yc), B.ptr(g),
                         xc.shape[0], n_mat, int(squared), wmin, wmax, B.ptr(gx), B.ptr(gy),
                         B.stream_of(xc))
        return gx.reshape(xs), gy.reshape(ys), None, None, None, None


class _SteinPdiv(torch.autograd.Function):
    """Pairwise Stein divergence over the rows [row_begin, row_end) of the pair list —
    PairwiseSteinDivergence (spd.py:246-295) + the value clamp / sqrt of stein_pdiv (191-194)."""

    @staticmethod
    def forward(ctx, x, n_mat, squared, wmin, row_begin, row_end):
        B.require_gpu(x)
        lib = B.lib()
        xc = x.detach().contiguous()
        n = xc.shape[0]
        dt = B.dtype_code(xc)
        npairs = B.pair_offset(n, row_end) - B.pair_offset(n, row_begin)
        ctx.save_for_backward(xc)
        ctx.empty = npairs == 0
        if ctx.empty:
            return xc.new_empty(0)
        with B.on_device(xc.device):
            ws = torch.empty(lib.raw('mm_spd_pdist_ws_bytes')(dt, n, n_mat), dtype=torch.uint8,
                             device=xc.device)
            out = torch.empty(npairs, dtype=xc.dtype, device=xc.device)
            lib.call('mm_spd_stein_pdiv_fwd', dt, B.ptr(xc), n, n_mat, row_begin, row_end, int(squared),
                     wmin, B.ptr(out), B.ptr(ws), 0, B.stream_of(xc))
        ctx.ws = ws
        ctx.args = (n_mat, squared, wmin, row_begin, row_end)
        return out

    @staticmethod
    def backward(ctx, g):
        xc, = ctx.saved_tensors
        if ctx.empty:
            return (torch.zeros_like(xc), ) + (None, ) * 5
        n_mat, squared, wmin, row_begin, row_end = ctx.args
        g = g.contiguous()
        with B.on_device(xc.device):
            grad = torch.empty_like(xc)
            B.lib().call('mm_spd_stein_pdiv_bwd', B.dtype_code(xc), B.ptr(xc), B.ptr(g), xc.shape[0], n_mat,
                         row_begin, row_end, int(squared), wmin, B.ptr(grad), B.ptr(ctx.ws),
                         B.MM_WS_PREPARED, B.stream_of(xc))
        return grad, None, None, None, None, None


class _SteinDiv(torch.autograd.Function):
    """Element-wise Stein divergence (spd.py:183-189)."""

    @staticmethod
    def forward(ctx, x, y, n_mat, squared, wmin):
        B.require_gpu(x, y)
        xc, yc = _flat(x.detach(), n_mat), _flat(y.detach(), n_mat)
        with B.on_device(xc.device):
            out = torch.empty(xc.shape[0], dtype=xc.dtype, device=xc.device)
            B.lib().call('mm_spd_stein_div', B.dtype_code(xc), B.ptr(xc), B.ptr(yc), None, xc.shape[0], n_mat,
                         int(squared), wmin, B.ptr(out), None, None, B.stream_of(xc))
        ctx.save_for_backward(xc, yc)
        ctx.args = (n_mat, squared, wmin, x.shape, y.shape)
        return out.reshape(x.shape[:-2])

    @staticmethod
    def backward(ctx, g):
        xc, yc = ctx.saved_tensors
        n_mat, squared, wmin, xs, ys = ctx.args
        g = g.reshape(-1).contiguous()
        with B.on_device(xc.device):
            gx, gy = torch.empty_like(xc), torch.empty_like(yc)
            B.lib().call('mm_spd_stein_div', B.dtype_code(xc), B.ptr(xc), B.ptr(yc), B.ptr(g), xc.shape[0],
                         n_mat, int(squared), wmin, None, B.ptr(gx), B.ptr(gy), B.stream_of(xc))
        return gx.reshape(xs), gy.reshape(ys), None, None, None


class SymmetricPositiveDefinite(Manifold):

    _warned_narrow = False

    def __init__(self, n, *, fast_symeig=True, fast_chol=True, use_stein_div=False, wmin=1e-8,
                 wmax=1e8, check_pd=False):
        self.n = n
        self.wmin = wmin
        self.wmax = wmax
        self.check_pd = check_pd
        self.use_stein_div = use_stein_div
        # Eigenvalue clamps narrower than [1e-6, 1e6] (never used by the reference's own scripts): the pair kernels' eigen-free
        # paths do not see eigenvalues and refuse such windows (csrc/spd_pair.hpp, spd_clamps_supported).  `pdist` of SPD(n >= 3)
        # then takes the element-wise kernels over the gathered pairs — what the reference's own Manifold.pdist does
        # (base.py:59-63) — and the fused objective / one-call step are off (the unfused composition runs instead).
        self.clamps_wide = wmin <= 1e-6 and wmax >= 1e6
        if not self.clamps_wide:
            self.pdist_loss = None
            if n >= 3 and not SymmetricPositiveDefinite._warned_narrow:
                SymmetricPositiveDefinite._warned_narrow = True   # once per process
                import warnings
                warnings.warn(f'SymmetricPositiveDefinite({n}, wmin={wmin:g}, wmax={wmax:g}): eigenvalue clamps narrower than '
                              '[1e-6, 1e6] are served exactly but OFF the pair kernels — pdist runs the element-wise kernels over '
                              'gathered pairs (chunked; several times slower) and the fused objective / one-call training step '
                              'are unavailable for this manifold (DESIGN.md section 6, item 11)', stacklevel=2)
        if use_stein_div:  # spd.py:51-53
            self.dist = self.stein_div
            self.pdist = self.stein_pdiv
            self.pdist_loss = None  # the fused objective is the affine-invariant one

    # -- Vec(.) of Pennec et al. (spd.py:66-81)
    @staticmethod
    def to_vec(x):
        n = x.shape[-1]
        fact = x.new_full((n, n), math.sqrt(2)).fill_diagonal_(1.0)
        return squareform0(fact * x)

    @staticmethod
    def from_vec(x_vec):
        # (always vector -> matrix: the reference's direction-guessing squareform0 mistakes a batch of
        # d(d+1)/2 vectors for ONE square matrix when the batch size equals d(d+1)/2 — SPD(2).rand(3) raises there)
        x = _to_square(x_vec / math.sqrt(2), nnp1d2_to_n(x_vec.shape[-1]), 0)
        x.diagonal(dim1=-2, dim2=-1).mul_(math.sqrt(2))
        return x

    @property
    def ndim(self):
        return 2

    @property
    def dim(self):
        return self.n * (self.n + 1) // 2

    def zero(self, *shape, out=None):
        return torch.eye(self.n, **_like(out)).repeat(*shape, 1, 1)

    def zero_vec(self, *shape, out=None):
        return torch.zeros(*shape, self.n, self.n, **_like(out))

    # -- per-point maps ------------------------------------------------------
    def _map(self, op, x, u=None):
        B.require_gpu(x, u)
        if torch.is_grad_enabled() and (x.requires_grad or (u is not None and u.requires_grad)):
            raise NotImplementedError(
                'SPD exp/log/retr/projx are optimizer-side maps (torch.no_grad); '
                'only dist/pdist are differentiable on the HIP path')
        shape = torch.broadcast_shapes(x.shape, u.shape) if u is not None else x.shape
        xc = _flat(x.expand(shape), self.n)
        uc = _flat(u.expand(shape), self.n) if u is not None else None
        with B.on_device(xc.device):
            out = torch.empty_like(xc)
            B.lib().call('mm_spd_map', B.dtype_code(xc), op, B.ptr(xc), B.ptr(uc), xc.shape[0],
                         self.n, self.wmin, self.wmax, B.ptr(out), B.stream_of(xc))
        return out.reshape(shape)

    def symeig(self, x):
        """Eigenvalues of sym(x), ascending (spd.py:35-41, 63-64; the reference's monitor reads them, monitor.py:39-45):
        one Jacobi eigensolve per matrix in registers (`mm_spd_eigvalsh`); batch shape kept.  Not differentiable
        (the differentiable path is `dist` / `pdist`); matrices wider than the kernels' range go to the GPU's
        `torch.linalg.eigvalsh`."""
        B.require_gpu(x)
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError('SPD.symeig is a monitoring map (no autograd on the HIP path): call it under '
                                      'torch.no_grad() or on x.detach(); dist / pdist are the differentiable entry points')
        if self.n > B.lib().raw('mm_spd_max_dim')():
            xd = x.detach()
            return torch.linalg.eigvalsh(0.5 * (xd + xd.transpose(-1, -2)))
        xc = _flat(x.detach(), self.n)
        out = torch.empty(xc.shape[0], self.n, dtype=xc.dtype, device=xc.device)
        if xc.shape[0]:
            with B.on_device(xc.device):
                B.lib().call('mm_spd_eigvalsh', B.dtype_code(xc), B.ptr(xc), xc.shape[0], self.n, B.ptr(out),
                             B.stream_of(xc))
        return out.reshape(*x.shape[:-2], self.n)

    def inner(self, x, u, v, keepdim=False):  # spd.py:96-106: tr(X^-1 U X^-1 V)
        assert not x.requires_grad and not u.requires_grad and not v.requires_grad
        p = self.norm(x, u + v, squared=True, keepdim=keepdim)
        q = self.norm(x, u - v, squared=True, keepdim=keepdim)
        return 0.25 * (p - q)

    def norm(self, x, u, squared=False, keepdim=False):  # spd.py:113-117
        B.require_gpu(x, u)
        shape = torch.broadcast_shapes(x.shape, u.shape)
        xc, uc = _flat(x.detach().expand(shape), self.n), _flat(u.detach().expand(shape), self.n)
        with B.on_device(xc.device):
            out = torch.empty(xc.shape[0], dtype=xc.dtype, device=xc.device)
            B.lib().call('mm_spd_norm', B.dtype_code(xc), B.ptr(xc), B.ptr(uc), xc.shape[0], self.n,
                         int(squared), B.ptr(out), B.stream_of(xc))
        out = out.reshape(shape[:-2])
        return out.reshape(*out.shape, 1, 1) if keepdim else out

    def proju(self, x, u, inplace=False):  # spd.py:119-124
        u_new = 0.5 * (u + u.transpose(-2, -1))
        if not inplace:
            return u_new
        u.set_(u_new)
        return u

    def projx(self, x, inplace=False):  # spd.py:126-132
        x_new = self._map(B.SPD_PROJX, x.detach() if inplace else x)
        if not inplace:
            return x_new
        x.set_(x_new)
        return x

    def rsgd_momentum_step(self, x, egrad, buf, *, lr, momentum, dampening, max_grad_norm=None, exact=False,
                           inplace=False):
        """Fused heavy-ball RiemannianSGD update (optim/rsgd.py:70-80; identity transport): `buf` is
        updated in place (kept symmetric); returns the new points, or None when not eligible."""
        d = self.n
        ok = (x.is_cuda and x.dtype in (torch.float32, torch.float64) and x.numel() > 0
              and d <= B.lib().raw('mm_spd_max_dim')() and buf.is_contiguous() and buf.dtype == x.dtype
              and buf.shape == x.shape)
        if not ok:
            return None
        xd = x.detach()
        inplace = inplace and xd.is_contiguous()
        xc = xd.reshape(-1, d, d).contiguous()
        gc = egrad.detach().reshape(-1, d, d).to(xc.dtype).contiguous()
        with B.on_device(xc.device):
            out = xc if inplace else torch.empty_like(xc)
            B.lib().call('mm_spd_rsgd_momentum_step', B.dtype_code(xc), B.ptr(xc), B.ptr(gc), B.ptr(buf),
                         xc.shape[0], d, float(lr), float(momentum), float(dampening),
                         -1.0 if max_grad_norm is None else float(max_grad_norm), int(bool(exact)), B.ptr(out),
                         B.stream_of(xc))
        return x if inplace else out.reshape(x.shape)

    def radam_step(self, x, egrad, exp_avg, exp_avg_sq, step, ticket, *, lr, betas, nc, eps, max_grad_norm=None,
                   exact=False, inplace=False):
        """Fused RiemannianAdam update (optim/radam.py:62-98) in one launch — see
        VectorManifold.radam_step; None when the tensors are not eligible."""
        d = self.n
        ok = (x.is_cuda and x.dtype in (torch.float32, torch.float64) and x.numel() > 0
              and d <= B.lib().raw('mm_spd_max_dim')() and exp_avg.is_contiguous() and exp_avg_sq.is_contiguous()
              and exp_avg.dtype == x.dtype and exp_avg_sq.dtype == x.dtype and exp_avg.shape == x.shape
              and exp_avg_sq.shape == x.shape)
        if not ok:
            return None
        xd = x.detach()
        inplace = inplace and xd.is_contiguous()
        xc = xd.reshape(-1, d, d).contiguous()
        gc = egrad.detach().reshape(-1, d, d).to(xc.dtype).contiguous()
        with B.on_device(xc.device):
            out = xc if inplace else torch.empty_like(xc)
            B.lib().call('mm_spd_radam_step', B.dtype_code(xc), B.ptr(xc), B.ptr(gc), B.ptr(exp_avg),
                         B.ptr(exp_avg_sq), B.ptr(step), B.ptr(ticket), xc.shape[0], d, float(lr), float(betas[0]),
                         float(betas[1] if betas[1] is not None else 0.0), int(bool(nc)), float(eps),
                         -1.0 if max_grad_norm is None else float(max_grad_norm), int(bool(exact)), B.ptr(out),
                         B.stream_of(xc))
        return x if inplace else out.reshape(x.shape)

    def egrad2rgrad(self, x, u):  # spd.py:134-135
        return self._map(B.SPD_EGRAD2RGRAD, x, u)

    def exp(self, x, u):  # spd.py:137-144
        return self._map(B.SPD_EXP, x, u)

    def retr(self, x, u):  # spd.py:146-154
        return self._map(B.SPD_RETR, x, u)

    def log(self, x, y):  # spd.py:156-161
        return self._map(B.SPD_LOG, x, y)

    # -- distances -------------------------------------------------------------
    def dist(self, x, y, squared=False, keepdim=False):  # spd.py:171-173
        shape = torch.broadcast_shapes(x.shape, y.shape)
        d = _SpdDist.apply(x.expand(shape), y.expand(shape), self.n, squared, self.wmin, self.wmax)
        return d.reshape(*d.shape, 1, 1) if keepdim else d

    def pdist(self, x, squared=False, rows=None):  # spd.py:175-181
        """All-pairs distances in row-major upper-triangle order.

        `rows=(row_begin,row_end)` restricts the result to the contiguous slice of the
        pair list owned by one shard (see graphembed.parallel)."""
        assert x.ndim == 3
        rb, re = (0, x.shape[0]) if rows is None else rows
        if not self.clamps_wide and self.n >= 3:   # (see __init__)
            return _SpdPdistGathered.apply(x, self.n, squared, self.wmin, self.wmax, int(rb), int(re))
        ext = B.autograd_ext()
        if ext is not None:   # the same two C-ABI calls as _SpdPdist, issued by C++ autograd nodes (csrc_torch/mm_autograd.cpp)
            try:
                return ext.spd_pdist(x, self.n, bool(squared), float(self.wmin), float(self.wmax), int(rb), int(re),
                                     bool(self.check_pd))
            except RuntimeError as e:
                if 'not positive-definite' in str(e):
                    raise torch.linalg.LinAlgError(str(e).split('linalg: ', 1)[-1]) from None
                if 'runs on MI355X only' in str(e) or 'failed:' in str(e):
                    raise B.BackendError(str(e)) from None
                raise
        return _SpdPdist.apply(x, self.n, squared, self.wmin, self.wmax, rb, re, self.check_pd)

    def pdist_loss(self, x, scale, target, spec, rows=None):
        """Fused `objective(target, softplus(scale) * pdist(x, squared=True))` with its gradients
        in one pass; `spec` comes from `objective_fn.fused_spec(epoch=, alpha=)`.  Returns the
        scalar loss of the pair range `rows` (all pairs by default)."""
        assert x.ndim == 3
        rb, re = (0, x.shape[0]) if rows is None else rows
        return _SpdPdistLoss.apply(x, scale, target, self.n, spec, self.wmin, self.wmax, rb, re)

    def stein_div(self, x, y, squared=False, keepdim=False):  # spd.py:183-189
        shape = torch.broadcast_shapes(x.shape, y.shape)
        d = _SteinDiv.apply(x.expand(shape), y.expand(shape), self.n, squared, self.wmin)
        return d.reshape(*d.shape, 1, 1) if keepdim else d

    def stein_pdiv(self, x, squared=False, rows=None):  # spd.py:191-194
        assert x.ndim == 3
        rb, re = (0, x.shape[0]) if rows is None else rows
        return _SteinPdiv.apply(x, self.n, squared, self.wmin, rb, re)

    def transp(self, x, y, u):  # spd.py:196-199
        return u

    def rsgd_step(self, x, egrad, *, lr, max_grad_norm=None, exact=False, inplace=False):
        """Fused momentum-free RiemannianSGD update (optim/rsgd.py:63-68,82):
        egrad2rgrad -> norm clip -> exp|retr in one kernel.  Returns the new points; with
        `inplace=True` they are written over `x` (every thread reads its point before writing it)."""
        B.require_gpu(x, egrad)
        xd = x.detach()
        inplace = inplace and xd.is_contiguous()
        xc, gc = _flat(xd, self.n), _flat(egrad.detach(), self.n)
        with B.on_device(xc.device):
            out = xc if inplace else torch.empty_like(xc)
            B.lib().call('mm_spd_rsgd_step', B.dtype_code(xc), B.ptr(xc), B.ptr(gc), xc.shape[0],
                         self.n, float(lr), -1.0 if max_grad_norm is None else float(max_grad_norm),
                         int(bool(exact)), B.ptr(out), B.stream_of(xc))
        return x if inplace else out.reshape(x.shape)

    def rand(self, *shape, out=None, ir=1e-1):  # spd.py:201-208
        eyes = self.zero(*shape, out=out)
        u = torch.randn(*shape, self.dim, dtype=eyes.dtype, device=eyes.device)
        u.div_(u.norm(dim=-1, keepdim=True)).mul_(ir)
        with torch.no_grad():
            return self.exp(eyes, self.from_vec(u))

    def randvec(self, x, norm=1):  # spd.py:210-221: X^1/2 U X^1/2
        shape = x.shape[:-2] + (self.dim, )
        u = torch.randn(shape, dtype=x.dtype, device=x.device)
        u.div_(u.norm(dim=-1, keepdim=True)).mul_(norm)
        u = self.from_vec(u)
        w, v = torch.linalg.eigh(x)
        xs = (v * w.clamp(self.wmin, self.wmax).sqrt().unsqueeze(-2)) @ v.transpose(-2, -1)
        return xs @ u @ xs.transpose(-2, -1)

    def __str__(self):
        return 'Manifold of {n}x{n} positive definite matrices'.format(n=self.n)
