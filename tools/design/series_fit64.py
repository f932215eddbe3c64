"""fp64 close-pair series of the SPD kernels: log(1+x) = x p(x) and log^2(1+x) = x^2 q(x) on |x| <= 0.3, by
Chebyshev interpolation carried out in extended precision (numpy longdouble, 64-bit mantissa) and converted to monomial
coefficients; prints the arrays for csrc/smallmat.hpp and the maximum error relative to |x| (resp. x^2) measured
against mpmath-free reference values (log1p in longdouble)."""
import numpy as np

LD = np.longdouble


def cheb_to_mono(c):
    """Chebyshev coefficients (in t) -> monomial coefficients (in t), longdouble."""
    n = len(c)
    T0 = np.zeros(n, LD); T0[0] = 1
    T1 = np.zeros(n, LD); T1[1] = 1
    out = c[0] * T0 + (c[1] * T1 if n > 1 else 0)
    for k in range(2, n):
        T2 = np.zeros(n, LD)
        T2[1:] = 2 * T1[:-1]
        T2 -= T0
        out = out + c[k] * T2
        T0, T1 = T1, T2
    return out


def fit(fun, r, n):
    k = np.arange(n, dtype=LD)
    t = np.cos((4 * np.arctan(LD(1))) * (k + LD(0.5)) / n)
    f = fun(LD(r) * t)
    c = np.array([(2 / LD(n)) * np.sum(f * np.cos((4 * np.arctan(LD(1))) * j * (k + LD(0.5)) / n)) for j in range(n)], LD)
    c[0] /= 2
    mono = cheb_to_mono(c) / LD(r) ** np.arange(n, dtype=LD)
    xs = np.linspace(-r, r, 20001).astype(LD)
    p = np.zeros_like(xs)
    for a in mono[::-1]:
        p = p * xs + a
    err = np.abs(p - fun(xs)).max()
    return mono, float(err)


def g1(x):
    small = np.abs(x) < 1e-4
    xs = np.where(small, LD(1), x)
    return np.where(small, 1 - x / 2 + x * x / 3 - x ** 3 / 4 + x ** 4 / 5, np.log1p(xs) / xs)


def g2(x):
    return g1(x) ** 2


if __name__ == '__main__':
    for name, fun, terms in (('log(1+x)/x', g1, (18, 20, 22)), ('log^2(1+x)/x^2', g2, (18, 20, 22))):
        for n in terms:
            c, e = fit(fun, 0.3, n)
            print(f'// {name}: {n} terms, max error {e:.2e}')
            print('  {' + ', '.join(f'{float(v):.17e}' for v in c) + '}')
