"""Compatibility shim that lets the 2019-era reference import under torch 2.10.

DEVELOPMENT-CONTAINER ONLY.  Used by ``gen_golden.py`` to import the real
reference from ``/root/reference`` (read-only) and record golden vectors.  It is
never imported by the product, by ``-m gpu`` tests, by ``smoke()`` or by
``bench.py``; ``/root/reference`` does not exist on the GPU box.

What it patches: a stub ``torch.utils.tensorboard`` (not
installed), and the APIs torch removed since 2019 — ``torch.symeig``,
``torch.solve``, ``torch.eig`` — re-expressed with ``torch.linalg``.
"""
import collections
import sys
import types

import torch

REFERENCE_ROOT = '/root/reference/graphembed'


def install():
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # the reference tree is read-only

    tb = types.ModuleType('torch.utils.tensorboard')

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None

    tb.SummaryWriter = SummaryWriter
    sys.modules['torch.utils.tensorboard'] = tb

    SE = collections.namedtuple('symeig', ['eigenvalues', 'eigenvectors'])

    def symeig(x, eigenvectors=False, upper=True):
        return SE(*torch.linalg.eigh(x, UPLO='U' if upper else 'L'))

    torch.symeig = symeig
    torch.Tensor.symeig = lambda self, eigenvectors=False, upper=True: symeig(
        self, eigenvectors, upper)

    SO = collections.namedtuple('solve', ['solution', 'LU'])
    torch.solve = lambda B, A: SO(torch.linalg.solve(A, B), None)

    EO = collections.namedtuple('eig', ['eigenvalues', 'eigenvectors'])

    def eig(x, eigenvectors=False):
        w = torch.linalg.eigvals(x)
        return EO(torch.stack([w.real, w.imag], -1), None)

    torch.eig = eig
    torch.Tensor.eig = lambda self, eigenvectors=False: eig(self, eigenvectors)
