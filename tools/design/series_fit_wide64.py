"""fp64 coefficients of the recentred series (smallmat.hpp, LogSeriesWide<double> / LogSqSeriesWide<double>):
log(1+x) = x p(x) and log^2(1+x) = x^2 q(x) on |x| <= 0.66, Chebyshev interpolation in 60-digit arithmetic (mpmath),
monomial coefficients rounded to double; prints the maximum error of the polynomial itself and of its evaluation by
Horner's rule in fp64.   python tools/design/series_fit_wide64.py [terms_p terms_q]"""
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60
R = mp.mpf('0.66')


def g1(x):
    return mp.log1p(x) / x if x != 0 else mp.mpf(1)


def g2(x):
    return g1(x) ** 2


def fit(fun, n):
    k = [mp.mpf(i) for i in range(n)]
    t = [mp.cos(mp.pi * (ki + mp.mpf('0.5')) / n) for ki in k]
    f = [fun(R * ti) for ti in t]
    c = [(mp.mpf(2) / n) * mp.fsum(fi * mp.cos(mp.pi * j * (ki + mp.mpf('0.5')) / n) for fi, ki in zip(f, k)) for j in range(n)]
    c[0] /= 2
    # Chebyshev -> monomial in t
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
    mono = [o / R ** i for i, o in enumerate(out)]
    return mono


def errors(mono, fun):
    xs = np.linspace(-0.66, 0.66, 4001)
    c64 = np.array([float(m) for m in mono])
    worst_poly, worst_eval = 0.0, 0.0
    for x in xs:
        if abs(x) < 1e-6:
            continue
        ref = fun(mp.mpf(float(x)))
        p = mp.mpf(0)
        for m in reversed(mono):
            p = p * mp.mpf(float(x)) + m
        worst_poly = max(worst_poly, abs(float(p - ref)))
        acc = 0.0
        for cc in c64[::-1]:
            acc = acc * x + cc
        worst_eval = max(worst_eval, abs(float(mp.mpf(acc) - ref)))
    return worst_poly, worst_eval


if __name__ == '__main__':
    if len(sys.argv) > 2:
        for name, fun, n in (('kA (log(1+x)/x)', g1, int(sys.argv[1])), ('kQ (log^2(1+x)/x^2)', g2, int(sys.argv[2]))):
            mono = fit(fun, n)
            print(name, n, 'terms; errors (poly, fp64 Horner):', errors(mono, fun))
            vals = [mp.nstr(m, 18) for m in mono]
            for i in range(0, n, 4):
                print('    ' + ', '.join(vals[i:i + 4]) + ',')
    else:
        for n in (30, 32, 34, 36, 38, 40):
            print(n, 'p:', errors(fit(g1, n), g1), ' q:', errors(fit(g2, n), g2))
