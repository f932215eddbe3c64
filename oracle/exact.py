"""TEST INFRASTRUCTURE ONLY — ctypes wrapper of oracle/exact.c (fp64, OpenMP)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, '_build', 'liboracle_exact.so')
_lib = None
KINDS = {'euclidean': 0, 'lorentz': 1, 'sphere': 2}


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_PATH):
            import subprocess
            subprocess.check_call(['make', '-C', _HERE])
        _lib = ctypes.CDLL(_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _chk(rc):
    if rc == -3:
        raise np.linalg.LinAlgError('input not positive definite')
    assert rc == 0, rc


def spd_pdist(x, squared=True, wmin=1e-8, wmax=1e8):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, d = x.shape[0], x.shape[-1]
    out = np.empty(n * (n - 1) // 2)
    _chk(lib().oracle_spd_pdist(_p(x), ctypes.c_long(n), d, int(squared), ctypes.c_double(wmin),
                                ctypes.c_double(wmax), _p(out)))
    return out


def spd_pdist_grad(x, g, squared=True, wmin=1e-8, wmax=1e8):
    x = np.ascontiguousarray(x, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    n, d = x.shape[0], x.shape[-1]
    grad = np.empty_like(x)
    _chk(lib().oracle_spd_pdist_grad(_p(x), _p(g), ctypes.c_long(n), d, int(squared), ctypes.c_double(wmin),
                                     ctypes.c_double(wmax), _p(grad)))
    return grad


def vec_pdist(kind, x, squared=True):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, m = x.shape
    out = np.empty(n * (n - 1) // 2)
    _chk(lib().oracle_vec_pdist(KINDS[kind], _p(x), ctypes.c_long(n), m, int(squared), _p(out)))
    return out


def vec_pdist_grad(kind, x, g, squared=True):
    x = np.ascontiguousarray(x, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    n, m = x.shape
    grad = np.empty_like(x)
    _chk(lib().oracle_vec_pdist_grad(KINDS[kind], _p(x), _p(g), ctypes.c_long(n), m, int(squared), _p(grad)))
    return grad
