"""Design script for the Cayley-transform matrix logarithm used by the SPD backward kernels:
log A = log(mu) I + 2 Z P(Z^2),  Z = (A - mu I)(A + mu I)^-1 = I - 2 mu (A + mu I)^-1,
P(w) ~ atanh(sqrt w)/sqrt w on [0, wmax] (near-minimax fit).  Prints coefficient tables and the
accuracy of a float32 emulation against an fp64 eigendecomposition."""
import sys
import numpy as np
from numpy.polynomial import chebyshev as C


def fit(wmax, K, npts=4001):
    # Chebyshev interpolation at Chebyshev nodes of f on [0,wmax] -> near-minimax
    k = np.arange(K + 1)
    t = np.cos(np.pi * (k + 0.5) / (K + 1))
    w = 0.5 * wmax * (t + 1)
    f = np.where(w > 1e-12, np.arctanh(np.sqrt(np.maximum(w, 1e-300))) / np.sqrt(np.maximum(w, 1e-300)), 1.0)
    cheb = C.chebfit(t, f, K)
    # convert to monomial in w:  t = 2w/wmax - 1
    p = C.cheb2poly(cheb)
    # compose
    from numpy.polynomial import polynomial as P
    lin = np.array([-1.0, 2.0 / wmax])
    out = np.zeros(1)
    powr = np.ones(1)
    for c in p:
        out = P.polyadd(out, c * powr)
        powr = P.polymul(powr, lin)
    ws = np.linspace(0, wmax, npts)
    fs = np.where(ws > 1e-12, np.arctanh(np.sqrt(np.maximum(ws, 1e-300))) / np.sqrt(np.maximum(ws, 1e-300)), 1.0)
    err = np.abs(P.polyval(ws, out) - fs).max()
    return out, err


if __name__ == '__main__':
    for wmax in (0.25, 0.36, 0.49):
        for K in range(4, 26):
            c, e = fit(wmax, K)
            print(f'wmax={wmax} K={K} maxerr={e:.2e}')
            if e < 1e-16:
                break


def emulate(a, coef, dt):
    """a: (N,3,3) SPD in dtype dt; float emulation of the device routine (same op order, no fma)."""
    f = dt
    a = a.astype(f)
    a00, a10, a11, a20, a21, a22 = a[:, 0, 0], a[:, 1, 0], a[:, 1, 1], a[:, 2, 0], a[:, 2, 1], a[:, 2, 2]
    mu = (a00 + a11 + a22) * f(1 / 3)
    b00, b11, b22 = a00 + mu, a11 + mu, a22 + mu
    b10, b20, b21 = a10, a20, a21
    c00 = b11 * b22 - b21 * b21
    c10 = b21 * b20 - b10 * b22
    c20 = b10 * b21 - b11 * b20
    c11 = b00 * b22 - b20 * b20
    c21 = b10 * b20 - b00 * b21
    c22 = b00 * b11 - b10 * b10
    det = b00 * c00 + b10 * c10 + b20 * c20
    rdet = f(1) / det
    # Z = (A - mu I) adj(B) / det  (commuting symmetric product) — relative accuracy also for A ~ mu I
    e = [a00 - mu, a10, a11 - mu, a20, a21, a22 - mu]
    adj = [c00, c10, c11, c20, c21, c22]
    e00, e10, e11, e20, e21, e22 = e
    z = [(e00 * c00 + e10 * c10 + e20 * c20) * rdet, (e10 * c00 + e11 * c10 + e21 * c20) * rdet,
         (e10 * c10 + e11 * c11 + e21 * c21) * rdet, (e20 * c00 + e21 * c10 + e22 * c20) * rdet,
         (e20 * c10 + e21 * c11 + e22 * c21) * rdet, (e20 * c20 + e21 * c21 + e22 * c22) * rdet]

    def sq(x):
        x00, x10, x11, x20, x21, x22 = x
        return [x00 * x00 + x10 * x10 + x20 * x20, x10 * x00 + x11 * x10 + x21 * x20,
                x10 * x10 + x11 * x11 + x21 * x21, x20 * x00 + x21 * x10 + x22 * x20,
                x20 * x10 + x21 * x11 + x22 * x21, x20 * x20 + x21 * x21 + x22 * x22]

    def mul(x, y):  # commuting symmetric product, packed
        x00, x10, x11, x20, x21, x22 = x
        y00, y10, y11, y20, y21, y22 = y
        return [x00 * y00 + x10 * y10 + x20 * y20, x10 * y00 + x11 * y10 + x21 * y20,
                x10 * y10 + x11 * y11 + x21 * y21, x20 * y00 + x21 * y10 + x22 * y20,
                x20 * y10 + x21 * y11 + x22 * y21, x20 * y20 + x21 * y21 + x22 * y22]
    w = sq(z)
    w2 = sq(w)
    t1 = w[0] + w[2] + w[5]
    trw2 = w2[0] + w2[2] + w2[5]
    t2 = f(0.5) * (t1 * t1 - trw2)
    t3 = (w[0] * (w[2] * w[5] - w[4] * w[4]) - w[1] * (w[1] * w[5] - w[4] * w[3]) + w[3] * (w[1] * w[4] - w[2] * w[3]))
    # P(W) = sum_k coef[k] W^k ; W^k = p I + q W + r W2
    c0 = np.full_like(mu, f(coef[0])); c1 = np.full_like(mu, f(coef[1])); c2 = np.full_like(mu, f(coef[2]))
    p, q, r = np.zeros_like(mu), np.zeros_like(mu), np.ones_like(mu)   # W^2
    for k in range(3, len(coef)):
        p, q, r = t3 * r, p - t2 * r, q + t1 * r
        c0 = c0 + f(coef[k]) * p; c1 = c1 + f(coef[k]) * q; c2 = c2 + f(coef[k]) * r
    pw = [c0 + c1 * w[0] + c2 * w2[0], c1 * w[1] + c2 * w2[1], c0 + c1 * w[2] + c2 * w2[2],
          c1 * w[3] + c2 * w2[3], c1 * w[4] + c2 * w2[4], c0 + c1 * w[5] + c2 * w2[5]]
    m = mul(z, pw)
    lm = np.log(mu)
    out = [lm + 2 * m[0], 2 * m[1], lm + 2 * m[2], 2 * m[3], 2 * m[4], lm + 2 * m[5]]
    L = np.zeros(a.shape, dtype=f)
    for (i, j), v in zip([(0, 0), (1, 0), (1, 1), (2, 0), (2, 1), (2, 2)], out):
        L[:, i, j] = v; L[:, j, i] = v
    return L, t1


def accuracy(dt, wmax, K, spread, N=200000, seed=0):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((N, 3, 3)))
    lam = rng.uniform(-spread, spread, (N, 3)) + rng.uniform(-0.5, 0.5, (N, 1))
    a64 = (q * np.exp(lam)[:, None, :]) @ q.transpose(0, 2, 1)
    a = a64.astype(dt).astype(np.float64)
    a = 0.5 * (a + a.transpose(0, 2, 1))
    wv, v = np.linalg.eigh(a)
    ref = (v * np.log(wv)[:, None, :]) @ v.transpose(0, 2, 1)
    coef, _ = fit(wmax, K)
    L, t1 = emulate(a, coef, dt)
    ok = t1 <= wmax
    err = np.abs(L.astype(np.float64) - ref).reshape(N, -1).max(1) / np.abs(ref).reshape(N, -1).max(1)
    return ok.mean(), err[ok].max(), np.median(err[ok])


if __name__ == '__main__' and len(sys.argv) > 1:
    for dt, K in ((np.float32, 6), (np.float32, 7), (np.float64, 13), (np.float64, 14)):
        for spread in (0.05, 0.3, 0.8, 1.2):
            print(dt.__name__, 'K', K, 'spread', spread, 'pass %.3f maxrel %.2e med %.2e' % accuracy(dt, 0.36, K, spread))
