"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — reference-faithful CPU port.

A torch-CPU restatement of graphembed's pairwise manifold-distance path.  It
keeps the reference's *operation sequence* (gather by ``triu_indices``,
per-pair ``L^-1 X L^-T`` einsum, closed-form eps-fudged 2x2 / 3x3 eigenvalues,
value-only clamps, autograd backward), so that its outputs — including the
bias the eps terms introduce — match the reference to rounding.  It is what
``bench.py`` times as ``cpu_baseline`` (kind "port").

All ``file:line`` citations are relative to
``/root/reference/graphembed/graphembed/``.

Pinned by ``tests/test_oracle_golden.py`` against vectors produced by the real
reference import (``tests/golden/gen_golden.py``).
"""
import math

import torch

EPS = 1e-8  # utils.py:13 — the same constant for fp32 and fp64


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
class _ValueClamp(torch.autograd.Function):
    """``t.data.clamp_(lo, hi)``: clamps the value, gradient is identity.

    The reference uses this idiom everywhere (e.g. spd.py:165,167;
    lorentz.py:74,76; sphere.py:70,72; base.py:31).
    """

    @staticmethod
    def forward(ctx, t, lo, hi):
        return t.clamp(min=lo, max=hi)

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def vclamp(t, lo=None, hi=None):
    return _ValueClamp.apply(t, lo, hi)


def triu_pairs(n, device=None):
    """Row-major (i<j) pair list — base.py:62, spd.py:179."""
    m = torch.triu_indices(n, n, 1, device=device)
    return m[0], m[1]


def sym(x):  # linalg/torch_batch.py:25-27
    return 0.5 * (x + x.transpose(-2, -1))


def axat(a, x):  # linalg/torch_batch.py:30-34
    return torch.einsum('...ij,...jk,...lk->...il', a, x, a)


def _eigh_upper(x):
    # the reference's torch.symeig(upper=True) (torch_batch.py:124-135)
    return torch.linalg.eigh(x, UPLO='U')


# --------------------------------------------------------------------------
# closed forms of linalg/fast.py
# --------------------------------------------------------------------------
def symeig2x2(X, eps=EPS):
    """fast.py:53-70 — reads X00, X11 and the *upper* off-diagonal only."""
    a, b, c = X[..., 0, 0], X[..., 1, 1], X[..., 0, 1]
    det = a * b - c**2
    ht = 0.5 * (a + b)
    delta = vclamp(ht**2 - det, eps)
    r = delta.sqrt()
    return torch.stack([ht - r, ht + r], dim=-1)


def det2x2(X):
    """fast.py:25-28 — the full matrix."""
    return X[..., 0, 0] * X[..., 1, 1] - X[..., 0, 1] * X[..., 1, 0]


def det3x3(X):
    """fast.py:31-37 — cofactor expansion along the first row, the full matrix."""
    m1 = X[..., 1, 1] * X[..., 2, 2] - X[..., 1, 2] * X[..., 2, 1]
    m2 = X[..., 1, 0] * X[..., 2, 2] - X[..., 1, 2] * X[..., 2, 0]
    m3 = X[..., 1, 0] * X[..., 2, 1] - X[..., 1, 1] * X[..., 2, 0]
    return X[..., 0, 0] * m1 - X[..., 0, 1] * m2 + X[..., 0, 2] * m3


def _symdet3x3(Y):
    """fast.py:40-50 — upper-triangle determinant of a symmetric 3x3."""
    y00, y01, y02 = Y[..., 0, 0], Y[..., 0, 1], Y[..., 0, 2]
    y11, y12, y22 = Y[..., 1, 1], Y[..., 1, 2], Y[..., 2, 2]
    return (y00 * y11 * y22 + 2 * y01 * y02 * y12 - y11 * y02**2 -
            y00 * y12**2 - y22 * y01**2)


symdet3x3 = _symdet3x3


def symeig3x3(X, eps=EPS):
    """fast.py:75-91 — trigonometric closed form with the +eps fudge terms."""
    batch = X.shape[:-2]
    q = (X.diagonal(dim1=-2, dim2=-1).sum(-1) / 3).reshape(*batch, 1, 1)
    Y = X - q * torch.eye(3, dtype=X.dtype, device=X.device)
    p = torch.sqrt(Y.pow(2).sum((-2, -1), keepdim=True) / 6)
    p.data.clamp_(min=eps)  # (in place, as fast.py:81: sqrt's backward then divides by the CLAMPED root)
    r = _symdet3x3(Y).reshape(*batch, 1, 1) / (2 * p.pow(3) + eps)
    r = vclamp(r, -1 + eps, 1 - eps)
    phi = torch.acos(r) / 3
    e1 = q + 2 * p * torch.cos(phi)
    e2 = q + 2 * p * torch.cos(phi + 2 * math.pi / 3)
    e3 = 3 * q - e1 - e2
    return torch.stack([e2, e3, e1], dim=-1).reshape(*batch, 3)


def _chol2x2_parts(X, eps=EPS):
    """fast.py:94-107 / 110-123: a, b, c of the eps-fudged 2x2 Cholesky."""
    shape = X.shape[:-2] + (1, 1)
    x00 = vclamp(X[..., 0, 0].reshape(shape), eps)
    x11 = X[..., 1, 1].reshape(shape)
    x01 = X[..., 0, 1].reshape(shape)
    a = x00.sqrt()
    b = x01 / a
    c = (x11 - b**2 + eps).sqrt()
    return a, b, c


def cholesky2x2(X, eps=EPS):
    a, b, c = _chol2x2_parts(X, eps)
    z = torch.zeros_like(a)
    return torch.cat([torch.cat([a, z], -1), torch.cat([b, c], -1)], -2)


def invcholesky2x2(X, ret_chol=False, eps=EPS):
    a, b, c = _chol2x2_parts(X, eps)
    det = vclamp(a * c, eps)
    z = torch.zeros_like(a)
    l_inv = torch.cat([torch.cat([c, z], -1), torch.cat([-b, a], -1)], -2) / det
    if not ret_chol:
        return l_inv, None
    return l_inv, torch.cat([torch.cat([a, z], -1), torch.cat([b, c], -1)], -2)


def singular_values_2x2(x, eps=EPS):
    """fast.py:138-159."""
    a, b, c, d = x[..., 0, 0], x[..., 0, 1], x[..., 1, 0], x[..., 1, 1]
    S1 = a**2 + b**2 + c**2 + d**2
    S2 = (a**2 + b**2 - c**2 - d**2)**2 + 4 * (a * c + b * d)**2
    S2 = torch.sqrt(vclamp(S2, eps))
    s1 = vclamp(0.5 * (S1 + S2), eps)
    s2 = vclamp(0.5 * (S1 - S2), eps)
    return torch.stack([torch.sqrt(s1), torch.sqrt(s2)], dim=-1)


# --------------------------------------------------------------------------
# manifolds
# --------------------------------------------------------------------------
class Manifold:
    """Default methods of manifolds/base.py:7-81."""

    ndim = 1

    def norm(self, x, u, squared=False, keepdim=False):  # base.py:29-33
        nsq = vclamp(self.inner(x, u, u, keepdim), EPS)
        return nsq if squared else nsq.sqrt()

    def egrad2rgrad(self, x, u):  # base.py:43-44
        return self.proju(x, u)

    def retr(self, x, u):  # base.py:49-50
        return self.exp(x, u)

    def dist(self, x, y, squared=False, keepdim=False):  # base.py:56-57
        return self.norm(x, self.log(x, y), squared, keepdim)

    def pdist(self, x, squared=False):  # base.py:59-63
        assert x.ndim == self.ndim + 1
        i, j = triu_pairs(x.shape[0], x.device)
        return self.dist(x[i], x[j], squared=squared)

    def transp(self, x, y, u):  # base.py:65-66
        return self.proju(y, u)


class SPD(Manifold):
    """manifolds/spd.py:21-243 (affine-invariant metric; Stein divergence: stein_div / stein_pdiv)."""

    ndim = 2

    def __init__(self, n, wmin=1e-8, wmax=1e8):
        self.n, self.wmin, self.wmax = n, wmin, wmax

    @property
    def dim(self):
        return self.n * (self.n + 1) // 2

    # -- spd.py:32-49 dispatch
    def symeig(self, x):
        if self.n == 2:
            return symeig2x2(x)
        if self.n == 3:
            return symeig3x3(x)
        return _eigh_upper(x)[0]

    def chol(self, x):
        return cholesky2x2(x) if self.n == 2 else torch.linalg.cholesky(x)

    def invchol(self, x, ret_chol=False):
        if self.n == 2:
            return invcholesky2x2(x, ret_chol)
        l = torch.linalg.cholesky(x)  # spd.py:55-61
        eye = torch.eye(self.n, dtype=x.dtype, device=x.device).expand_as(l)
        l_inv = torch.linalg.solve_triangular(l, eye, upper=False)
        return l_inv, (l if ret_chol else None)

    # -- spd.py:66-81
    @staticmethod
    def from_vec(v):
        m = v.shape[-1]
        n = int(math.floor(math.sqrt(2 * m)))
        iu = torch.triu_indices(n, n)
        x = torch.zeros(*v.shape[:-1], n, n, dtype=v.dtype)
        off = v / math.sqrt(2)
        x[..., iu[0], iu[1]] = off
        x[..., iu[1], iu[0]] = off
        d = torch.arange(n)
        x[..., d, d] = x[..., d, d] * math.sqrt(2)
        return x

    def zero(self, *shape, dtype=None):
        return torch.eye(self.n, dtype=dtype).repeat(*shape, 1, 1)

    def _lult(self, x, u, ret_chol=False):  # spd.py:108-111
        l_inv, l = self.invchol(x, ret_chol)
        return axat(l_inv, u), l

    def norm(self, x, u, squared=False, keepdim=False):  # spd.py:113-117
        lult, _ = self._lult(x, u)
        nsq = lult.pow(2).sum((-2, -1), keepdim=keepdim)
        return nsq if squared else nsq.sqrt()

    def proju(self, x, u):  # spd.py:119-124
        return sym(u)

    def projx(self, x):  # spd.py:126-132 + torch_batch.py:145-153
        w, v = _eigh_upper(sym(x))
        w = w.clamp(self.wmin, self.wmax)
        return torch.einsum('...ij,...j,...kj->...ik', v, w, v)

    def egrad2rgrad(self, x, u):  # spd.py:134-135
        return axat(x, sym(u))

    def exp(self, x, u):  # spd.py:137-144
        lult, l = self._lult(x, u, ret_chol=True)
        w, v = _eigh_upper(lult)
        return axat(l, torch.einsum('...ij,...j,...kj->...ik', v, w.exp(), v))

    def retr(self, x, u):  # spd.py:146-154
        l = self.chol(x)
        lu = torch.linalg.solve_triangular(l, u, upper=False)
        return sym(x + u + 0.5 * lu.transpose(-2, -1) @ lu)

    def log(self, x, y):  # spd.py:156-161
        lylt, l = self._lult(x, y, ret_chol=True)
        w, v = _eigh_upper(lylt)
        return axat(l, torch.einsum('...ij,...j,...kj->...ik', v, w.log(), v))

    def _norm_log(self, a, squared=False, keepdim=False):  # spd.py:163-169
        w = vclamp(self.symeig(a), self.wmin, self.wmax)
        dsq = vclamp(w.log().pow(2).sum(-1, keepdim=keepdim), self.wmin)
        return dsq if squared else dsq.sqrt()

    def dist(self, x, y, squared=False, keepdim=False):  # spd.py:171-173
        lylt, _ = self._lult(x, y)
        return self._norm_log(lylt, squared, keepdim)

    def pdist(self, x, squared=False):  # spd.py:175-181
        assert x.ndim == 3
        l_inv, _ = self.invchol(x)
        i, j = triu_pairs(x.shape[0], x.device)
        return self._norm_log(axat(l_inv[i], x[j]), squared)

    # -- Stein divergence (spd.py:183-194, 246-295; linalg/torch_batch.py:173-197) -------------------
    def _logdet(self, x):
        """2 sum log |diag chol(x)| — PLogDet.forward; autograd of this expression yields g X^-1."""
        return 2 * self.chol(x).diagonal(dim1=-2, dim2=-1).abs().log().sum(-1)

    def stein_div(self, x, y, squared=False, keepdim=False):
        div = self._logdet(0.5 * (x + y)) - 0.5 * (self._logdet(x) + self._logdet(y))
        div = vclamp(div, self.wmin)
        div = div if squared else div.sqrt()
        return div.reshape(*div.shape, 1, 1) if keepdim else div

    def stein_pdiv(self, x, squared=False):
        assert x.ndim == 3
        i, j = triu_pairs(x.shape[0], x.device)
        ld = self._logdet(x)
        div = self._logdet(0.5 * (x[i] + x[j])) - 0.5 * (ld[i] + ld[j])
        div = vclamp(div, self.wmin)
        return div if squared else div.sqrt()

    def transp(self, x, y, u):  # spd.py:196-199
        return u

    def rand(self, n, ir=1e-1, dtype=None, generator=None):  # spd.py:201-208
        u = torch.randn(n, self.dim, dtype=dtype, generator=generator)
        u = u / u.norm(dim=-1, keepdim=True) * ir
        return self.exp(self.zero(n, dtype=u.dtype), self.from_vec(u))


def ldot(u, v, keepdim=False):
    """Minkowski inner product — lorentz.py:101-122 (plain autograd here)."""
    # same summation order as the reference: negate the time term in place,
    # then one sum over all coordinates (matters in fp32 when -<x,y> ~ 1)
    uv = u * v
    uv = torch.cat([-uv[..., :1], uv[..., 1:]], dim=-1)
    return uv.sum(-1, keepdim=keepdim)


class _Acosh(torch.autograd.Function):
    """lorentz.py:125-138: backward divides by max(sqrt(x^2-1), EPS)."""

    @staticmethod
    def forward(ctx, x):
        z = torch.sqrt(x * x - 1)
        ctx.save_for_backward(z)
        return torch.log(x + z)

    @staticmethod
    def backward(ctx, g):
        z, = ctx.saved_tensors
        return g / z.clamp(min=EPS)


class Lorentz(Manifold):
    """manifolds/lorentz.py:9-98."""

    ndim = 1

    def __init__(self, n):
        self.n = n

    def inner(self, x, u, v, keepdim=False):
        return ldot(u, v, keepdim)

    def proju(self, x, u):  # lorentz.py:39-42
        return u + ldot(x, u, keepdim=True) * x

    def projx(self, x):  # lorentz.py:44-50
        t = torch.sqrt(1 + x[..., 1:].pow(2).sum(-1, keepdim=True))
        return torch.cat([t, x[..., 1:]], dim=-1)

    def egrad2rgrad(self, x, u):  # lorentz.py:52-57
        u = torch.cat([-u[..., :1], u[..., 1:]], dim=-1)
        return self.proju(x, u)

    def exp(self, x, u):  # lorentz.py:59-62
        un = ldot(u, u, keepdim=True).clamp(min=0).sqrt().clamp(min=EPS)
        return x * un.cosh() + un.sinh() * u / un

    def log(self, x, y):  # lorentz.py:64-70
        xy = ldot(x, y, keepdim=True).clamp(max=-1)
        denom = torch.sqrt(xy * xy - 1).clamp(min=EPS)
        num = _Acosh.apply(-xy).clamp(min=EPS)
        return self.proju(x, num / denom * (y + xy * x))

    def dist(self, x, y, squared=False, keepdim=False):  # lorentz.py:72-77
        d = vclamp(-ldot(x, y), 1)
        dist = vclamp(_Acosh.apply(d), EPS)
        return dist.pow(2) if squared else dist

    def transp(self, x, y, u):  # lorentz.py:79-82
        xy = ldot(x, y, keepdim=True)
        uy = ldot(u, y, keepdim=True)
        return u + uy / (1 - xy) * (x + y)

    def rand(self, n, ir=1e-2, dtype=None, generator=None):  # lorentz.py:84-86
        x = torch.empty(n, self.n, dtype=dtype).uniform_(-ir, ir, generator=generator)
        return self.projx(x)


class Sphere(Manifold):
    """manifolds/sphere.py:8-99 (vector case)."""

    ndim = 1

    def __init__(self, n):
        self.n = n

    def inner(self, x, u, v, keepdim=False):
        return (u * v).sum(-1, keepdim=keepdim)

    def proju(self, x, u):  # sphere.py:41-44
        return u - self.inner(None, x, u, keepdim=True) * x

    def projx(self, x):  # sphere.py:46-49
        return x / self.norm(None, x, keepdim=True)

    def exp(self, x, u):  # sphere.py:51-56
        nu = self.norm(None, u, keepdim=True)
        e = x * torch.cos(nu) + u * torch.sin(nu) / nu
        return torch.where(nu > EPS, e, self.retr(x, u))

    def retr(self, x, u):  # sphere.py:58-59
        return self.projx(x + u)

    def log(self, x, y):  # sphere.py:61-66
        u = self.proju(x, y - x)
        d = self.dist(x, y, keepdim=True)
        return torch.where(d > EPS, u * d / self.norm(None, u, keepdim=True), u)

    def dist(self, x, y, squared=False, keepdim=False):  # sphere.py:68-74
        c = vclamp(self.inner(None, x, y, keepdim), -1 + EPS**2, 1 - EPS**2)
        th = vclamp(torch.acos(c), EPS)
        return th.pow(2) if squared else th

    def rand(self, n, ir=1e-2, dtype=None, generator=None):  # sphere.py:76-79,91-96
        x = torch.zeros(n, self.n, dtype=dtype)
        x[..., 0] = -1
        u = self.proju(x, torch.randn(n, self.n, dtype=dtype, generator=generator))
        u = u / u.norm(dim=-1, keepdim=True) * ir
        return self.retr(x, u)


class Euclidean(Manifold):
    """manifolds/euclidean.py:7-60 (vector case)."""

    ndim = 1

    def __init__(self, n):
        self.n = n

    def inner(self, x, u, v, keepdim=False):
        return (u * v).sum(-1, keepdim=keepdim)

    def proju(self, x, u):
        return u

    def projx(self, x):
        return x

    def exp(self, x, u):
        return x + u

    def log(self, x, y):
        return y - x

    def rand(self, n, ir=1e-2, dtype=None, generator=None):
        return torch.empty(n, self.n, dtype=dtype).uniform_(-ir, ir, generator=generator)


class Grassmann(Manifold):
    """manifolds/grassmann.py:10-116 (retr='svd' default)."""

    ndim = 2

    def __init__(self, n, p):
        self.n, self.p = n, p

    def inner(self, x, u, v, keepdim=False):
        return (u * v).sum((-2, -1), keepdim=keepdim)

    def proju(self, x, u):  # grassmann.py:49-53
        return u - x @ (x.transpose(-2, -1) @ u)

    def projx(self, x):  # grassmann.py:55-61
        return torch.linalg.qr(x)[0]

    def exp(self, x, u):  # grassmann.py:63-69
        us, ss, vh = torch.linalg.svd(u, full_matrices=False)
        vs = vh.transpose(-2, -1)
        lhs = x @ torch.einsum('...ij,...j,...kj->...ik', vs, ss.cos(), vs)
        return lhs + torch.einsum('...ij,...j,...kj->...ik', us, ss.sin(), vs)

    def retr(self, x, u):  # grassmann.py:76-80
        uu, _, vh = torch.linalg.svd(x + u, full_matrices=False)
        return uu @ vh

    def log(self, x, y):  # grassmann.py:82-89
        ytx = y.transpose(-2, -1) @ x
        At = y.transpose(-2, -1) - ytx @ x.transpose(-2, -1)
        Bt = torch.linalg.solve(ytx, At)
        us, ss, vh = torch.linalg.svd(Bt.transpose(-2, -1), full_matrices=False)
        vs = vh.transpose(-2, -1)
        return torch.einsum('...ij,...j,...kj->...ik', us[..., :self.p],
                            ss[..., :self.p].atan(), vs[..., :self.p])

    def dist(self, x, y, squared=False, keepdim=False):  # grassmann.py:91-96
        xty = x.transpose(-2, -1) @ y
        if self.p == 2:
            s = singular_values_2x2(xty)
        else:
            s = torch.linalg.svdvals(xty)
        s = vclamp(s, -1 + EPS**2, 1 - EPS**2)
        dsq = s.acos().pow(2).sum(-1, keepdim=keepdim)
        return dsq if squared else dsq.sqrt()


class Stiefel(Manifold):
    """manifolds/stiefel.py:7-93 — projections / retractions only."""

    ndim = 2

    def __init__(self, n, p):
        self.n, self.p = n, p

    def inner(self, x, u, v, keepdim=False):
        return (u * v).sum((-2, -1), keepdim=keepdim)

    def proju(self, x, u):  # stiefel.py:40-45
        return u - x @ sym(x.transpose(-2, -1) @ u)

    def orthonormalize(self, x):  # stiefel.py:47-50
        q, r = torch.linalg.qr(x)
        return q * r.diagonal(dim1=-2, dim2=-1).sign().unsqueeze(-2)

    def retr(self, x, u):  # stiefel.py:66-69 (svd / polar)
        uu, _, vh = torch.linalg.svd(x + u, full_matrices=False)
        return uu @ vh

    def retr_qr(self, x, u):  # stiefel.py:62-63
        return self.orthonormalize(x + u)


# --------------------------------------------------------------------------
# callers on the path
# --------------------------------------------------------------------------
def rsgd_step(man, x, grad, *, lr, momentum=0.0, dampening=0.0,
              max_grad_norm=None, exact=False, momentum_buffer=None):
    """One RiemannianSGD update of one parameter — optim/rsgd.py:40-82.

    Returns ``(new_x, new_momentum_buffer_or_None)``.
    """
    with torch.no_grad():
        if momentum > 0 and momentum_buffer is None:
            momentum_buffer = grad.clone()  # rsgd.py:53-54
        step = man.exp if exact else man.retr
        g = man.egrad2rgrad(x, grad)
        if max_grad_norm is not None:  # rsgd.py:66-68
            gn = man.norm(x, g, keepdim=True)
            g = g * torch.clamp(max_grad_norm / gn, max=1.0)
        if momentum > 0:  # rsgd.py:71-78
            buf = momentum_buffer * momentum + (1 - dampening) * g
            new_x = step(x, -lr * buf)
            return new_x, man.transp(x, new_x, buf)
        return step(x, -lr * g), None  # rsgd.py:82


def compute_dists(mans, xs, scales, idx=None):
    """ManifoldEmbedding.compute_dists — modules.py:84-88."""
    sp = torch.nn.functional.softplus
    return sum(sp(s) * m.pdist(x if idx is None else x[idx], squared=True)
               for m, x, s in zip(mans, xs, scales))


def stress_loss(gd, md):  # objectives.py:39-45
    return (md - gd).pow(2).sum()


def quotient_loss(gd, md, *, epoch, alpha, inc_l1=True, inc_l2=True):  # objectives.py:16-36
    gd = gd * alpha
    loss = 0
    if inc_l1:
        loss = loss + (md / gd - 1.0).abs().sum()
    if inc_l2:
        loss = loss + (gd / (md + 1.0 / (epoch + 1)) - 1.0).abs().sum()
    return loss


def make(name, *args):
    return {'spd': SPD, 'lorentz': Lorentz, 'sphere': Sphere,
            'euclidean': Euclidean, 'grassmann': Grassmann,
            'stiefel': Stiefel}[name](*args)


# ---------------------------------------------------------------- metrics.py:61-96
def mean_average_precision(dense, neighbours):
    """py_mean_average_precision restated on plain numpy: `dense` is the (n, n) matrix of embedding
    distances (zero diagonal), `neighbours[u]` the set of graph neighbours of node u."""
    import numpy as np
    n = dense.shape[0]
    scores = []
    for u in range(n):
        order = np.argsort(dense[u], kind='stable')
        nb = neighbours[u]
        hits, psum = 0, 0.0
        for i in range(1, n):
            if order[i] in nb:
                hits += 1
                psum += hits / i
                if hits == len(nb):
                    break
        scores.append(psum / len(nb))
    return float(np.mean(scores))


# ---------------------------------------------------------------- pyx/impl/precision.cpp:300-429
def layer_f1_scores(dense, hops, per_tree_average=False, min_degree=1, max_degree=10**9, degrees=None):
    """LayerMeanF1Scores / LayerMeanAverageF1Scores restated on numpy (unweighted graphs): `dense` the
    (n, n) embedding distances, `hops` the (n, n) hop distances (= layers of every shortest-path tree).
    Walks each tree's nodes in embedding order exactly as the C++ does (ordered multiset -> counts)."""
    import numpy as np
    n = dense.shape[0]
    width = int(hops.max())
    m1, m2, cnt = np.zeros(width), np.zeros(width), np.zeros(width)
    for u in range(n):
        if degrees is not None and not (min_degree <= degrees[u] <= max_degree):
            continue
        order = [v for v in np.argsort(dense[u], kind='stable') if v != u]
        layer_sizes = np.bincount(hops[u], minlength=width + 1)
        strict_before = np.concatenate([[0, 0], np.cumsum(layer_sizes[1:])])  # nodes on layers 1..l-1
        seen = np.zeros(width + 1, dtype=np.int64)
        t1, t2, tc = np.zeros(width), np.zeros(width), np.zeros(width)
        for i, v in enumerate(order, start=1):
            l = hops[u, v]
            nodes_before = seen[:l + 1].sum() + 1
            precision = nodes_before / i
            recall = nodes_before / (strict_before[l] + seen[l] + 1)
            f1 = 2 * precision * recall / (precision + recall)
            seen[l] += 1
            t1[l - 1] += f1
            t2[l - 1] += f1 * f1
            tc[l - 1] += 1
        if per_tree_average:
            has = tc > 0
            mean = np.where(has, t1 / np.maximum(tc, 1), 0.0)
            m1 += mean
            m2 += mean * mean
            cnt += has
        else:
            m1 += t1
            m2 += t2
            cnt += tc
    with np.errstate(invalid='ignore', divide='ignore'):  # layers nobody contributed to -> nan, as in the C++
        means = m1 / cnt
        return means, m2 / cnt - means * means
