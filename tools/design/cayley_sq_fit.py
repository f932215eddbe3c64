"""Design script for the invariants-only squared distance of a pair outside the close-pair gate
(smallmat.hpp, logsq_cayley3): with mu = 2^k next to tr A / 3, E = A - mu I (eigenvalues eps_k) and
z_k = eps_k / (2 mu + eps_k) = (lambda_k - mu) / (lambda_k + mu),

    sum_k log^2 lambda_k = 2 log(mu) log det A - 3 log^2(mu) + 4 sum_k atanh^2(z_k),

atanh^2(sqrt w) = w q(w), and sum_k w_k q(w_k) = tr(W q(W)) from the elementary symmetric functions of the w_k = z_k^2
alone — which are rational in the invariants (s1, s2, s3) of E:

    D = 8 mu^3 + 4 mu^2 s1 + 2 mu s2 + s3 = det(A + mu I),
    e1(z) = (4 mu^2 s1 + 4 mu s2 + 3 s3) / D,  e2(z) = (2 mu s2 + 3 s3) / D,  e3(z) = s3 / D.

No matrix inverse, no matrix product; log det A comes from the per-node table (log det X_j - log det X_i).
Prints q's monomial coefficients (Chebyshev interpolation on [0, wmax] in 60-digit arithmetic) and the accuracy of an
fp64 / fp32 emulation of the device routine against an mpmath evaluation.
    python tools/design/cayley_sq_fit.py [K64 K32]"""
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60
WMAX = mp.mpf('0.36')


def q_of(w):
    if w == 0:
        return mp.mpf(1)
    r = mp.sqrt(w)
    return (mp.atanh(r) / r) ** 2


def fit(n, fun=q_of, wmax=WMAX):
    """n coefficients (degree n - 1), monomial in w."""
    t = [mp.cos(mp.pi * (mp.mpf(i) + mp.mpf('0.5')) / n) for i in range(n)]
    f = [fun(wmax * (ti + 1) / 2) for ti in t]
    c = [(mp.mpf(2) / n) * mp.fsum(fi * mp.cos(mp.pi * j * (mp.mpf(i) + mp.mpf('0.5')) / n) for i, fi in enumerate(f))
         for j in range(n)]
    c[0] /= 2
    # Chebyshev series in t -> monomial in t -> monomial in w (t = 2 w / wmax - 1)
    T0 = [mp.mpf(0)] * n; T0[0] = mp.mpf(1)
    T1 = [mp.mpf(0)] * n
    if n > 1:
        T1[1] = mp.mpf(1)
    out = [c[0] * a + (c[1] * b if n > 1 else 0) for a, b in zip(T0, T1)]
    for j in range(2, n):
        T2 = [mp.mpf(0)] + [2 * v for v in T1[:-1]]
        T2 = [a - b for a, b in zip(T2, T0)]
        out = [o + c[j] * v for o, v in zip(out, T2)]
        T0, T1 = T1, T2
    # compose with t = a w + b
    a, b = 2 / wmax, mp.mpf(-1)
    mono = [mp.mpf(0)] * n
    powr = [mp.mpf(1)] + [mp.mpf(0)] * (n - 1)     # (a w + b)^k as a polynomial in w
    for k in range(n):
        for i in range(n):
            mono[i] += out[k] * powr[i]
        nxt = [mp.mpf(0)] * n
        for i in range(n):
            nxt[i] += b * powr[i]
            if i + 1 < n:
                nxt[i + 1] += a * powr[i]
        powr = nxt
    return mono


def poly_error(mono, fun=q_of, wmax=WMAX, npts=2001):
    worst = mp.mpf(0)
    for i in range(npts):
        w = wmax * i / (npts - 1)
        p = mp.mpf(0)
        for m in reversed(mono):
            p = p * w + m
        worst = max(worst, abs(p - fun(w)) / fun(w))
    return float(worst)


def emulate(a, coef, dt):
    """a: (N,3,3) SPD, already rounded to dt; returns (d2, tr Z^2) with the device routine's op order."""
    f = dt
    a = a.astype(f)
    a00, a10, a11, a20, a21, a22 = a[:, 0, 0], a[:, 1, 0], a[:, 1, 1], a[:, 2, 0], a[:, 2, 1], a[:, 2, 2]
    mean = (a00 + a11 + a22) * f(1 / 3)
    mant, k = np.frexp(mean)
    k = np.where(mant < 0.70710678118654752, k - 1, k)
    mu = np.ldexp(f(1), k).astype(f)
    lmu = (k * 0.69314718055994531).astype(f)
    e00, e11, e22 = a00 - mu, a11 - mu, a22 - mu
    s1 = e00 + e11 + e22
    p2 = e00 * e00 + e11 * e11 + e22 * e22 + f(2) * (a10 * a10 + a20 * a20 + a21 * a21)
    s2 = f(0.5) * (s1 * s1 - p2)
    s3 = e00 * (e11 * e22 - a21 * a21) - a10 * (a10 * e22 - a21 * a20) + a20 * (a10 * a21 - e11 * a20)
    m2 = mu + mu
    D = ((m2 + s1) * m2 + s2) * m2 + s3
    r = f(1) / D
    n1 = ((m2 * s1) + f(2) * s2) * m2 + f(3) * s3
    n2 = m2 * s2 + f(3) * s3
    z1, z2, z3 = n1 * r, n2 * r, s3 * r
    t1 = z1 * z1 - f(2) * z2
    t2 = z2 * z2 - f(2) * z1 * z3
    t3 = z3 * z3
    p1 = t1
    pp2 = t1 * p1 - f(2) * t2
    pp3 = t1 * pp2 - t2 * p1 + f(3) * t3
    K = len(coef)
    c0 = np.full_like(mu, f(coef[K - 3])); c1 = np.full_like(mu, f(coef[K - 2])); c2 = np.full_like(mu, f(coef[K - 1]))
    for i in range(K - 4, -1, -1):
        c0, c1, c2 = c2 * t3 + f(coef[i]), c0 - c2 * t2, c1 + c2 * t1
    S = c0 * p1 + c1 * pp2 + c2 * pp3
    return lmu, f(4) * S, t1


def accuracy(dt, K, spread, N=4000, seed=0, shift=0.5):
    rng = np.random.default_rng(seed)
    qm, _ = np.linalg.qr(rng.standard_normal((N, 3, 3)))
    lam = rng.uniform(-spread, spread, (N, 3)) + rng.uniform(-shift, shift, (N, 1))
    a64 = (qm * np.exp(lam)[:, None, :]) @ qm.transpose(0, 2, 1)
    a = a64.astype(dt).astype(np.float64)
    a = 0.5 * (a + a.transpose(0, 2, 1))
    coef = [float(c) for c in fit(K)]
    lmu, s4, t1 = emulate(a, coef, dt)
    ok = t1 <= float(WMAX)
    worst = 0.0
    for i in range(N):
        if not ok[i]:
            continue
        w = mp.eigsy(mp.matrix(a[i].tolist()), eigvals_only=True)
        ref = mp.fsum(mp.log(x) ** 2 for x in w)
        ld = mp.fsum(mp.log(x) for x in w)
        ldr = dt(float(ld))    # the table's log det, rounded
        got = dt(2) * lmu[i] * ldr - dt(3) * lmu[i] * lmu[i] + s4[i]
        worst = max(worst, float(abs(mp.mpf(float(got)) - ref) / (mp.mpf('1e-2') + ref)))
    return ok.mean(), worst


if __name__ == '__main__':
    if len(sys.argv) > 2:
        for name, K in (('fp64', int(sys.argv[1])), ('fp32', int(sys.argv[2]))):
            mono = fit(K)
            print(name, K, 'coefficients, relative error of the polynomial', poly_error(mono))
            vals = [mp.nstr(m, 18) for m in mono]
            for i in range(0, K, 4):
                print('    ' + ', '.join(vals[i:i + 4]) + ',')
        for dt, K in ((np.float64, int(sys.argv[1])), (np.float32, int(sys.argv[2]))):
            for spread in (0.05, 0.3, 0.8, 1.2):
                print(dt.__name__, 'K', K, 'spread', spread, 'inside the gate %.3f, max |err| / (1e-2 + d2) %.2e' % accuracy(dt, K, spread))
    else:
        for K in range(6, 20):
            print(K, 'coefficients: relative error', poly_error(fit(K)))
