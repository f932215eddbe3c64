"""Economised (Chebyshev-interpolated) polynomials for the close-pair series of the SPD kernels:
log(1+x) = x * p(x) and log^2(1+x) = x^2 * q(x) on |x| <= 0.3 (the close-pair gate).  Prints monomial
coefficients and the maximum error relative to |x| (resp. x^2)."""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P


def fit(fun, r, n):
    k = np.arange(n); t = np.cos(np.pi * (k + 0.5) / n); x = r * t
    ch = C.chebfit(t, fun(x), n - 1)
    mono = C.cheb2poly(ch)  # in t = x / r
    mono = mono / r ** np.arange(n)
    xs = np.linspace(-r, r, 40001)
    err = np.abs(P.polyval(xs, mono) - fun(xs)).max()
    return mono, err


def g1(x):
    x = np.where(np.abs(x) < 1e-9, 1e-9, x)
    return np.log1p(x) / x


def g2(x):
    x = np.where(np.abs(x) < 1e-9, 1e-9, x)
    return (np.log1p(x) / x) ** 2


if __name__ == '__main__':
    for name, fun in (('log(1+x)/x', g1), ('log^2(1+x)/x^2', g2)):
        for n in (7, 8, 9, 10):
            c, e = fit(fun, 0.3, n)
            print(name, 'terms', n, 'err %.2e' % e)
            print('   ', ', '.join('%.9ef' % v for v in c))
