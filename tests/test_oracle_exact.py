"""Pins the plain-C fp64 checker (oracle/exact.c) against the reference-faithful port in fp64,
which is itself pinned against the reference's golden vectors.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sym
from oracle import exact
from oracle import ref_port as rp


@pytest.mark.parametrize('d', [2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize('init', ['rand', 'wide'])
def test_spd_against_reference_golden(d, init):
    G = load_golden(f'spd{d}')
    tag = f'f64/{init}/n33'
    x, g = G[f'{tag}/x'], G[f'{tag}/g']
    # the reference's eps-fudged closed forms (d = 2, 3) bias its fp64 results by up to ~1e-6
    tol = 2e-6 if d <= 3 else 1e-9
    for sq, key in ((True, 'd2'), (False, 'd1')):
        out = exact.spd_pdist(x, squared=sq)
        np.testing.assert_allclose(out * out if not sq else out, G[f'{tag}/{key}'] ** (1 if sq else 2), rtol=tol, atol=1e-7 if d <= 3 else 1e-12)
    gr = exact.spd_pdist_grad(x, g, squared=True)
    ref = sym(G[f'{tag}/grad_d2'])
    assert np.abs(gr - ref).max() <= (5e-6 if d <= 3 else 1e-9) * np.abs(ref).max()
    assert np.abs(gr - np.swapaxes(gr, 1, 2)).max() <= 1e-12 * np.abs(gr).max()


@pytest.mark.parametrize('key,kind,m', [('lorentz11', 'lorentz', 11), ('sphere6', 'sphere', 6), ('euclidean10', 'euclidean', 10),
                                          ('lorentz48', 'lorentz', 48), ('sphere64', 'sphere', 64), ('euclidean40', 'euclidean', 40)])
@pytest.mark.parametrize('init', ['rand', 'wide'])
def test_vec_against_reference_golden(key, kind, m, init):
    G = load_golden(key)
    tag = f'f64/{init}/n33'
    x, g = G[f'{tag}/x'], G[f'{tag}/g']
    np.testing.assert_allclose(exact.vec_pdist(kind, x, True), G[f'{tag}/d2'], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(exact.vec_pdist(kind, x, False), G[f'{tag}/d1'], rtol=1e-9, atol=1e-13)
    for sq, gk in ((True, 'grad_d2'), (False, 'grad_d1')):
        gr = exact.vec_pdist_grad(kind, x, g, sq)
        assert np.abs(gr - G[f'{tag}/{gk}']).max() <= 1e-8 * np.abs(G[f'{tag}/{gk}']).max()


def test_spd4_against_port_seeded():
    gen = torch.Generator().manual_seed(0)
    port = rp.SPD(4)
    x = port.rand(150, ir=1.0, dtype=torch.float64, generator=gen)
    g = torch.randn(150 * 149 // 2, dtype=torch.float64, generator=gen)
    xr = x.clone().requires_grad_()
    d2 = port.pdist(xr, squared=True)
    gr, = torch.autograd.grad((d2 * g).sum(), xr)
    np.testing.assert_allclose(exact.spd_pdist(x.numpy()), d2.detach().numpy(), rtol=1e-10, atol=1e-13)
    ref = sym(gr.numpy())
    assert np.abs(exact.spd_pdist_grad(x.numpy(), g.numpy()) - ref).max() <= 1e-9 * np.abs(ref).max()
    with pytest.raises(np.linalg.LinAlgError):
        exact.spd_pdist(-np.eye(3)[None].repeat(3, 0))
