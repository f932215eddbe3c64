"""Design script for log_cayley3_ring (smallmat.hpp): the Cayley-transform logarithm of a 3x3 SPD matrix with every step
but E^2 and the last combination carried out on SCALARS, in the quotient rings R[Z]/(chi_Z) and R[E]/(chi_E).

  mu = 2^k next to tr A / 3,  E = A / mu - I  (exact scaling),  s1, s2, s3 = elementary symmetric functions of E's spectrum,
  Z = E (E + 2 I)^-1 = (s3 I + 2 (s1 + 2) E - 2 E^2) / D,   D = det(E + 2 I) = 8 + 4 s1 + 2 s2 + s3
      (adj(B) = B^2 - tr(B) B + e2(B) I and Cayley-Hamilton for E),
  e_i(Z) rational in (s1, s2, s3) (tools/design/cayley_sq_fit.py), e_i(W = Z^2) from those,
  atanh(Z) = Z P(W):  P(W) = c0 + c1 W + c2 W^2 by Horner in R[W]/(chi_W)  (as log_cayley3),
           = beta0 + beta1 Z + beta2 Z^2    (c0 Z + c1 Z^3 + c2 Z^5 reduced in R[Z]/(chi_Z): three steps of three FMAs),
           = gamma0 + gamma1 E + gamma2 E^2 (Z and Z^2 written in the E basis: scalar ring arithmetic),
  log A = log(mu) I + 2 (gamma0 I + gamma1 E + gamma2 E^2).

Matrix work: E^2 (18 multiply-adds) and the last line (12) against adjugate + three commuting products + P(W) assembly
(~105) in log_cayley3.  This script emulates the routine in fp64 / fp32 and compares with an mpmath eigendecomposition."""
import sys

import mpmath as mp
import numpy as np

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from cayley_fit import fit as fit_p   # noqa: E402

mp.mp.dps = 40


def emulate(a, coef, dt):
    f = dt
    a = a.astype(f)
    a00, a10, a11, a20, a21, a22 = a[:, 0, 0], a[:, 1, 0], a[:, 1, 1], a[:, 2, 0], a[:, 2, 1], a[:, 2, 2]
    mean = (a00 + a11 + a22) * f(1 / 3)
    mant, k = np.frexp(mean)
    k = np.where(mant < 0.70710678118654752, k - 1, k)
    r = np.ldexp(f(1), -k).astype(f)
    lmu = (k * 0.69314718055994531).astype(f)
    e00, e11, e22 = a00 * r - f(1), a11 * r - f(1), a22 * r - f(1)
    e10, e20, e21 = a10 * r, a20 * r, a21 * r
    f00 = e00 * e00 + e10 * e10 + e20 * e20
    f11 = e10 * e10 + e11 * e11 + e21 * e21
    f22 = e20 * e20 + e21 * e21 + e22 * e22
    f10 = e10 * e00 + e11 * e10 + e21 * e20
    f20 = e20 * e00 + e21 * e10 + e22 * e20
    f21 = e20 * e10 + e21 * e11 + e22 * e21
    s1 = e00 + e11 + e22
    s2 = f(0.5) * (s1 * s1 - (f00 + f11 + f22))
    s3 = e00 * (e11 * e22 - e21 * e21) - e10 * (e10 * e22 - e21 * e20) + e20 * (e10 * e21 - e11 * e20)
    D = f(8) + f(4) * s1 + f(2) * s2 + s3
    rD = f(1) / D
    z1 = (f(4) * (s1 + s2) + f(3) * s3) * rD
    z2 = (f(2) * s2 + f(3) * s3) * rD
    z3 = s3 * rD
    t1 = z1 * z1 - f(2) * z2
    t2 = z2 * z2 - f(2) * z1 * z3
    t3 = z3 * z3
    K = len(coef)
    c0 = np.full_like(r, f(coef[K - 3])); c1 = np.full_like(r, f(coef[K - 2])); c2 = np.full_like(r, f(coef[K - 1]))
    for i in range(K - 4, -1, -1):
        c0, c1, c2 = c2 * t3 + f(coef[i]), c0 - c2 * t2, c1 + c2 * t1
    # Z (c2 Z^4 + c1 Z^2 + c0) in R[Z]/(chi_Z), Z^3 = z1 Z^2 - z2 Z + z3
    b0, b1, b2 = c1, np.zeros_like(r), c2                       # c2 Z^2 + c1
    b0, b1, b2 = b2 * z3, b0 - b2 * z2, b1 + b2 * z1            # . Z
    b0, b1, b2 = b2 * z3 + c0, b0 - b2 * z2, b1 + b2 * z1       # . Z + c0
    b0, b1, b2 = b2 * z3, b0 - b2 * z2, b1 + b2 * z1            # . Z
    # Z and Z^2 in the E basis
    y0, y1, y2 = z3, f(2) * (s1 + f(2)) * rD, -f(2) * rD
    q0, q1, q2, q3, q4 = y0 * y0, f(2) * y0 * y1, f(2) * y0 * y2 + y1 * y1, f(2) * y1 * y2, y2 * y2
    h2, h1, h0 = s1 * s1 - s2, s3 - s1 * s2, s1 * s3            # E^4 = h2 E^2 + h1 E + h0
    w0 = q0 + q3 * s3 + q4 * h0
    w1 = q1 - q3 * s2 + q4 * h1
    w2 = q2 + q3 * s1 + q4 * h2
    g0 = b0 + b1 * y0 + b2 * w0
    g1 = b1 * y1 + b2 * w1
    g2 = b1 * y2 + b2 * w2
    g0 = f(2) * g0 + lmu
    g1 = f(2) * g1
    g2 = f(2) * g2
    out = [g0 + g1 * e00 + g2 * f00, g1 * e10 + g2 * f10, g0 + g1 * e11 + g2 * f11,
           g1 * e20 + g2 * f20, g1 * e21 + g2 * f21, g0 + g1 * e22 + g2 * f22]
    L = np.zeros(a.shape, dtype=f)
    for (i, j), v in zip([(0, 0), (1, 0), (1, 1), (2, 0), (2, 1), (2, 2)], out):
        L[:, i, j] = v; L[:, j, i] = v
    return L, t1


def accuracy(dt, K, spread, N=3000, seed=0, shift=0.5, wmax=0.36):
    rng = np.random.default_rng(seed)
    qm, _ = np.linalg.qr(rng.standard_normal((N, 3, 3)))
    lam = rng.uniform(-spread, spread, (N, 3)) + rng.uniform(-shift, shift, (N, 1))
    a64 = (qm * np.exp(lam)[:, None, :]) @ qm.transpose(0, 2, 1)
    a = a64.astype(dt).astype(np.float64)
    a = 0.5 * (a + a.transpose(0, 2, 1))
    coef, _ = fit_p(wmax, K)
    L, t1 = emulate(a, coef, dt)
    ok = t1 <= wmax
    worst, errs = 0.0, []
    for i in range(N):
        if not ok[i]:
            continue
        w, v = mp.eigsy(mp.matrix(a[i].tolist()))
        ref = v * mp.diag([mp.log(x) for x in w]) * v.T
        scale = max(abs(ref[r, c]) for r in range(3) for c in range(3))
        err = max(abs(mp.mpf(float(L[i, r, c])) - ref[r, c]) for r in range(3) for c in range(3)) / (mp.mpf('1e-3') + scale)
        errs.append(float(err))
    return ok.mean(), max(errs), float(np.median(errs))


if __name__ == '__main__':
    for dt, K in ((np.float64, 13), (np.float32, 6)):
        for spread, shift in ((0.05, 0.0), (0.05, 0.5), (0.3, 0.5), (0.8, 0.5), (1.2, 0.5), (0.5, 8.0)):
            print(dt.__name__, 'K', K, 'spread', spread, 'shift', shift,
                  'inside the gate %.3f, max err / (1e-3 + max|log A|) %.2e, median %.2e' % accuracy(dt, K, spread, shift=shift))
