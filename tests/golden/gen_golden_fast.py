"""Golden vectors of graphembed.linalg.fast from the REAL reference (development container only).
    PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0 python tests/golden/gen_golden_fast.py
Every function of linalg/fast.py:25-159, fp32 + fp64: inputs, the function's `eps`, the reference's outputs and the
gradient its autograd returns for a recorded cotangent.  Inputs are the reference's own test fixtures
(tests/conftest.py:27-43: `rand_sym`, `rand_spd`; tests/test_linalg.py:47-141) plus the cases its guards exist for
(identity matrices, a rank-one 2x2, a tiny leading entry, non-symmetric input that shows which half is read, a custom
eps).  Output: tests/golden/fast.npz (data only)."""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from graphembed.linalg import fast  # noqa: E402  (the reference's)
from gen_golden import np_, DT  # noqa: E402


def rand_sym(n, d):  # tests/conftest.py:27-33
    x = torch.rand(n, d, d)
    return 0.5 * (x + x.transpose(1, 2))


def rand_spd(n, d):  # tests/conftest.py:36-43
    x = torch.rand(n, d, d)
    return (x @ x.transpose(1, 2)).add_(torch.eye(d))


def near_identity(n, d):  # the SPD manifold's initialisation range (spd.py:183-190): I + small symmetric noise
    e = 1e-2 * torch.randn(n, d, d)
    return torch.eye(d) + 0.5 * (e + e.transpose(1, 2))


def inputs(fn):
    n = 37
    if fn in ('symeig2x2', 'det2x2'):
        d = 2
    elif fn in ('symeig3x3', 'det3x3', 'symdet3x3'):
        d = 3
    else:
        d = 2
    cases = {}
    if fn.startswith('symeig'):
        cases['rand_sym'] = rand_sym(n, d)
        cases['rand_spd'] = rand_spd(n, d)
        cases['near_identity'] = near_identity(n, d)
        cases['identity'] = torch.eye(d).expand(5, -1, -1).clone()
        cases['nonsym'] = torch.rand(n, d, d) + torch.eye(d)
    elif fn in ('cholesky2x2', 'invcholesky2x2', 'invcholesky2x2_chol'):
        cases['rand_spd'] = rand_spd(n, 2)
        cases['near_identity'] = near_identity(n, 2)
        tiny = rand_spd(6, 2)
        tiny[:, 0, 0] = 1e-10          # the clamp of x00 is active
        tiny[:, 0, 1] = tiny[:, 1, 0] = 0.0
        cases['tiny_x00'] = tiny
        cases['nonsym'] = rand_spd(n, 2) + 0.1 * torch.rand(n, 2, 2)
    elif fn == 'singular_values_2x2':
        cases['rand'] = torch.rand(n, 2, 2)
        cases['randn'] = torch.randn(n, 2, 2)
        u, v = torch.randn(9, 2, 1), torch.randn(9, 1, 2)
        cases['rank_one'] = u @ v
    else:
        cases['rand'] = torch.rand(n, d, d)
        cases['randn'] = torch.randn(n, d, d)
    return cases


def call(fn, x, eps):
    if fn == 'invcholesky2x2':
        return (fast.invcholesky2x2(x, ret_chol=False, eps=eps)[0], )
    if fn == 'invcholesky2x2_chol':
        return fast.invcholesky2x2(x, ret_chol=True, eps=eps)
    if fn in ('det2x2', 'det3x3', 'symdet3x3'):
        return (getattr(fast, fn)(x), )
    return (getattr(fast, fn)(x, eps=eps), )


def main():
    out = {}
    fns = ['det2x2', 'det3x3', 'symdet3x3', 'symeig2x2', 'symeig3x3', 'cholesky2x2', 'invcholesky2x2',
           'invcholesky2x2_chol', 'singular_values_2x2']
    for fn in fns:
        for dname in DT:
            torch.set_default_dtype(DT[dname])
            torch.manual_seed(zlib.crc32(repr((fn, dname, 'fast')).encode()) % (2**31))
            for case, x0 in inputs(fn).items():
                for eps in ((1e-8, 1e-4) if case in ('rand_spd', 'rand') and not fn.startswith('det') and fn != 'symdet3x3'
                            else (1e-8, )):
                    key = f'{fn}/{dname}/{case}/eps{eps:g}'
                    x = x0.clone().requires_grad_()
                    # (cholesky2x2 / invcholesky2x2 clamp x00 of their ARGUMENT in place, fast.py:97-98: hand them a copy)
                    ys = call(fn, x.clone() if 'cholesky' in fn else x, eps)
                    gs = [torch.randn_like(y) for y in ys]
                    loss = sum((y * g).sum() for y, g in zip(ys, gs))
                    loss.backward()
                    out[key + '/x'] = np_(x0)
                    for k, (y, g) in enumerate(zip(ys, gs)):
                        out[key + f'/out{k}'] = np_(y)
                        out[key + f'/cot{k}'] = np_(g)
                    out[key + '/grad'] = np_(x.grad)
    np.savez_compressed(os.path.join(HERE, 'fast.npz'), **out)
    print(len(out), 'arrays')
    torch.set_default_dtype(torch.float32)


if __name__ == '__main__':
    main()
