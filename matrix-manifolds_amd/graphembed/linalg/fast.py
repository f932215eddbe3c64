"""`graphembed.linalg.fast` on the gfx950 kernels of csrc/fast.hip — the names and call signatures of
graphembed/graphembed/linalg/fast.py:25-159 (`det2x2`, `det3x3`, `symdet3x3`, `symeig2x2`, `symeig3x3`, `cholesky2x2`,
`invcholesky2x2`, `singular_values_2x2`), which the reference's manifolds select (spd.py:35-46, grassmann.py:29) and
which its monitor, tests/test_linalg.py:47-141 and tests/test_perf.py:14-81 call directly.

Every function is differentiable and returns what the reference returns for the same input: the forward is fast.py's
arithmetic (same half of the symmetric matrix read, same `eps` guards), the backward is the derivative torch's autograd
derives from it — gradients on the upper triangle only, as fast.py:5-10 warns; callers symmetrise.  One launch per call in
either direction (`mm_fast_fwd` / `mm_fast_bwd`); GPU tensors only, no CPU arithmetic in this package.  The backward is a raw
kernel launch, not a differentiable torch expression: it is marked `once_differentiable`, so a double backward
(`create_graph=True`) raises instead of silently returning gradients without a graph (the reference's pure-torch fast.py is
twice differentiable; nothing on the training path differentiates twice).

CPU tensors: laid over a maintainer's checkout (INTEGRATION.md option A) this module SHADOWS the checkout's
`graphembed/linalg/fast.py`, whose CPU callers (monitor.py, tests/test_linalg.py, the analysis scripts) would break — so a
call with a CPU tensor is handed to the checkout's own function of the same name when a checkout is on the path (the
reference serving the reference's callers: nothing of it is shipped or copied), and raises as before when there is none.

`graphembed.linalg` is a namespace package (the reference's has no `__init__.py` either): over a maintainer's checkout
(INTEGRATION.md option A) `graphembed.linalg.fast` is this module and `graphembed.linalg.torch_batch` the checkout's.

Deviation: the reference's `cholesky2x2` / `invcholesky2x2` clamp `X[..., 0, 0]` of the ARGUMENT in place (`x00` is a view
and `x00.data.clamp_` writes through it, fast.py:97-98, 113-114); here the argument is never written.
"""
import functools
import importlib.util
import os
import sys

import torch
from torch.autograd.function import once_differentiable

from graphembed import _backend as B

_reference = []   # [module or None] once looked up


def _reference_fast():
    """The checkout's linalg/fast.py (loaded once under a private name), or None without a checkout behind this package."""
    if not _reference:
        mod = None
        from graphembed import _overlay
        for base in _overlay.later_packages():
            path = os.path.join(base, 'linalg', 'fast.py')
            if os.path.isfile(path):
                name = 'graphembed.linalg._overlaid_fast'
                spec = importlib.util.spec_from_file_location(name, path)
                mod = importlib.util.module_from_spec(spec)
                mod.__package__ = 'graphembed.linalg'
                sys.modules[name] = mod
                try:
                    spec.loader.exec_module(mod)
                except BaseException:
                    sys.modules.pop(name, None)
                    raise
                break
        _reference.append(mod)
    return _reference[0]


def _cpu_to_checkout(fn):
    """CPU tensors go to the checkout's function of the same name when there is a checkout (see the module docstring)."""
    @functools.wraps(fn)
    def wrapper(X, *args, **kwargs):
        if isinstance(X, torch.Tensor) and not X.is_cuda:
            ref = _reference_fast()
            if ref is not None:
                return getattr(ref, fn.__name__)(X, *args, **kwargs)
        return fn(X, *args, **kwargs)
    return wrapper


def _flat(x, k):
    assert x.shape[-2:] == (k, k), f'expected [..., {k}, {k}], got {tuple(x.shape)}'
    B.require_gpu(x)
    return x.detach().reshape(-1, k, k).contiguous()


class _Fast(torch.autograd.Function):
    """out (and the optional second output) of one `mm_fast_fwd`; `mm_fast_bwd` on the way back."""

    @staticmethod
    def forward(ctx, x, op, k, out_shape, out2_shape, eps):
        xc = _flat(x, k)
        n = xc.shape[0]
        with B.on_device(xc.device):
            out = torch.empty((n, ) + out_shape, dtype=xc.dtype, device=xc.device)
            out2 = torch.empty((n, ) + out2_shape, dtype=xc.dtype, device=xc.device) if out2_shape is not None else None
            if n:
                B.lib().call('mm_fast_fwd', op, B.dtype_code(xc), B.ptr(xc), n, float(eps), B.ptr(out), B.ptr(out2),
                             B.stream_of(xc))
        ctx.save_for_backward(xc)
        ctx.args = (op, float(eps), x.shape)
        batch = x.shape[:-2]
        out = out.reshape(batch + out_shape)
        if out2 is None:
            return out
        return out, out2.reshape(batch + out2_shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, g, g2=None):
        xc, = ctx.saved_tensors
        op, eps, shape = ctx.args
        n = xc.shape[0]
        with B.on_device(xc.device):
            gx = torch.empty_like(xc)
            if n:
                g = g.reshape(n, -1).contiguous()
                g2 = g2.reshape(n, -1).contiguous() if g2 is not None else None
                B.lib().call('mm_fast_bwd', op, B.dtype_code(xc), B.ptr(xc), B.ptr(g), B.ptr(g2), n, eps, B.ptr(gx),
                             B.stream_of(xc))
        return gx.reshape(shape), None, None, None, None, None


def _det(x, op, k, keepdim):
    det = _Fast.apply(x, op, k, (), None, 0.0)
    return det.view(-1, 1, 1) if keepdim else det


@_cpu_to_checkout
def det2x2(X, keepdim=False):  # fast.py:25-28
    return _det(X, B.FAST_DET2, 2, keepdim)


@_cpu_to_checkout
def det3x3(X, keepdim=False):  # fast.py:31-37
    return _det(X, B.FAST_DET3, 3, keepdim)


@_cpu_to_checkout
def symdet3x3(X, keepdim=False):  # fast.py:40-50 (upper triangle)
    return _det(X, B.FAST_SYMDET3, 3, keepdim)


@_cpu_to_checkout
def symeig2x2(X, eps=1e-8):
    """Eigenvalues of symmetric 2x2 matrices, ascending (fast.py:53-70); reads x00, x11, x01."""
    return _Fast.apply(X, B.FAST_SYMEIG2, 2, (2, ), None, eps)


@_cpu_to_checkout
def symeig3x3(X, eps=1e-8):
    """Eigenvalues of symmetric 3x3 matrices, ascending, by the trigonometric formula (fast.py:75-91); squeezed like the
    reference's return value."""
    return _Fast.apply(X, B.FAST_SYMEIG3, 3, (3, ), None, eps).squeeze()


@_cpu_to_checkout
def cholesky2x2(X, eps=1e-8):
    """Lower Cholesky factor of 2x2 SPD matrices (fast.py:94-107)."""
    return _Fast.apply(X, B.FAST_CHOLESKY2, 2, (2, 2), None, eps)


@_cpu_to_checkout
def invcholesky2x2(X, ret_chol=False, eps=1e-8):
    """(L^-1, L or None) of 2x2 SPD matrices (fast.py:110-135)."""
    if not ret_chol:
        return _Fast.apply(X, B.FAST_INVCHOLESKY2, 2, (2, 2), None, eps), None
    return _Fast.apply(X, B.FAST_INVCHOLESKY2, 2, (2, 2), (2, 2), eps)


@_cpu_to_checkout
def singular_values_2x2(x, eps=1e-8):
    """Singular values of 2x2 matrices, descending (fast.py:138-159)."""
    return _Fast.apply(x, B.FAST_SINGULAR2, 2, (2, ), None, eps)
