"""Economised polynomial of log(1+x)/x on |x| <= R for the RECENTRED close-pair series of the SPD backward
(smallmat.hpp, log_series3_centred): log A = log(mu) I + log(I + E'), E' = A/mu - I with mu = tr A / d, so E' is
traceless and its spectral radius is at most sqrt((d-1)/d) ||E'||_F.  Prints monomial coefficients (fp32) and the
maximum error relative to |x|, in exact arithmetic and evaluated in fp32 by Horner's rule."""
import sys

import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P


def g1(x):
    x = np.where(np.abs(x) < 1e-12, 1e-12, x)
    return np.log1p(x) / x


def fit(r, n):
    k = np.arange(n)
    t = np.cos(np.pi * (k + 0.5) / n)
    ch = C.chebfit(t, g1(r * t), n - 1)
    mono = C.cheb2poly(ch) / r ** np.arange(n)
    xs = np.linspace(-r, r, 200001)
    err = np.abs(P.polyval(xs, mono) - g1(xs)).max()
    c32 = mono.astype(np.float32)
    acc = np.zeros_like(xs, dtype=np.float32)
    x32 = xs.astype(np.float32)
    for c in c32[::-1]:
        acc = acc * x32 + c
    err32 = np.abs(acc.astype(np.float64) - g1(x32.astype(np.float64))).max()
    return mono, err, err32


if __name__ == '__main__':
    r = float(sys.argv[1]) if len(sys.argv) > 1 else 0.66
    for n in range(12, 22):
        c, e, e32 = fit(r, n)
        print(f'R = {r} terms {n}: max err {e:.2e} (fp32 Horner {e32:.2e})')
        if len(sys.argv) > 2 and int(sys.argv[2]) == n:
            print('   ', ', '.join('%.9ef' % v for v in c))


def g2(x):
    return g1(x) ** 2


def fit2(r, n):
    """log^2(1+x) = x^2 q(x): the forward's recentred series (logsq_series3_centred)"""
    k = np.arange(n)
    t = np.cos(np.pi * (k + 0.5) / n)
    ch = C.chebfit(t, g2(r * t), n - 1)
    mono = C.cheb2poly(ch) / r ** np.arange(n)
    xs = np.linspace(-r, r, 200001)
    err = np.abs(P.polyval(xs, mono) - g2(xs)).max()
    return mono, err


if __name__ == '__main__' and len(sys.argv) > 3 and sys.argv[3] == 'sq':
    r = float(sys.argv[1])
    for n in range(12, 22):
        c, e = fit2(r, n)
        print(f'q: R = {r} terms {n}: max err {e:.2e}')
        if int(sys.argv[2]) == n:
            print('   ', ', '.join('%.9ef' % v for v in c))
