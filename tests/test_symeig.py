"""SymmetricPositiveDefinite.symeig (spd.py:35-41, 63-64): golden vectors recorded from the real reference
(tests/golden/gen_golden_symeig.py) against the oracle port on the CPU and against `mm_spd_eigvalsh` on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'symeig.npz'))
DT = {'f32': torch.float32, 'f64': torch.float64}


@pytest.mark.parametrize('d', range(2, 10))
@pytest.mark.parametrize('dname', list(DT))
@pytest.mark.parametrize('init', ['rand', 'wide'])
def test_oracle_port_symeig_matches_reference(d, dname, init):
    """The port restates the reference's closed forms (n = 2, 3) and its LAPACK route: same values."""
    from oracle import ref_port as rp
    x = torch.from_numpy(GOLD[f'spd{d}/{dname}/{init}/x'])
    w = torch.from_numpy(GOLD[f'spd{d}/{dname}/{init}/w']).double()
    got = torch.sort(rp.SPD(d).symeig(x), dim=-1).values.double()
    tol = 2e-5 if dname == 'f32' else 1e-11
    assert (got - w).abs().max().item() <= tol * w.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize('d', range(2, 10))
@pytest.mark.parametrize('dname', list(DT))
@pytest.mark.parametrize('init', ['rand', 'wide'])
def test_symeig_gpu(d, dname, init):
    """One Jacobi eigensolve per matrix.  Against the reference's values to ITS accuracy — its closed forms for n = 2, 3
    are off by up to 2e-5 (fp32) / 8e-6 (fp64, n = 3) of the largest eigenvalue at near-degenerate spectra (X ~ I: the
    reference's own initialisation) — and against numpy's eigvalsh of the fp64 input to rounding."""
    from graphembed.manifolds import SymmetricPositiveDefinite as SPD
    x = torch.from_numpy(GOLD[f'spd{d}/{dname}/{init}/x'])
    w = GOLD[f'spd{d}/{dname}/{init}/w'].astype(np.float64)
    got = SPD(d).symeig(x.cuda())
    assert got.shape == (x.shape[0], d) and got.dtype == x.dtype
    got = got.double().cpu().numpy()
    assert (np.diff(got, axis=-1) >= 0).all(), 'ascending'
    scale = np.abs(w).max()
    ref_tol = 5e-5 if dname == 'f32' else (2e-5 if d == 3 else 1e-11)
    assert np.abs(got - w).max() <= ref_tol * scale
    xs = x.double().numpy()
    exact = np.linalg.eigvalsh(0.5 * (xs + xs.transpose(0, 2, 1)))
    assert np.abs(got - exact).max() <= (2e-6 if dname == 'f32' else 1e-13) * scale
    # batch shape is kept; an empty batch is fine
    assert SPD(d).symeig(x.cuda().reshape(2, -1, d, d)).shape == (2, x.shape[0] // 2, d)
    assert SPD(d).symeig(x.cuda()[:0]).shape == (0, d)
