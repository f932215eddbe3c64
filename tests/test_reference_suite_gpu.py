"""The reference's own small test files, re-expressed on this package's classes with their sizes and tolerances (atol 1e-4,
graphembed/tests/helpers/utils.py): tests/test_isometry.py (SPD(2) of fixed determinant <-> the hyperboloid of curvature -1/2,
sphere <-> hyperboloid near the pole), tests/test_euclidean.py, tests/test_sphere.py, and the closed-form checks of
tests/test_linalg.py that have a counterpart here (symeig 2x2 / 3x3).  (tests/test_spd.py: test_spd_gpu.py;
tests/test_optim.py: test_vec_gpu.py; tests/test_metrics.py: test_metrics.py; tests/test_ortho.py: test_mat_gpu.py.)"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ATOL = 1e-4
F64 = dict(dtype=torch.float64, device='cuda')


def close(a, b, atol=ATOL):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=0)


def ldot(u, v):
    return (u[..., 1:] * v[..., 1:]).sum(-1) - u[..., 0] * v[..., 0]


def unit_det_spd2_to_hyperboloid(x):
    """[[a, c], [c, b]] -> ((a+b)/2, (a-b)/2, c), scaled onto <y, y>_L = -1."""
    a, b, c = x[..., 0, 0], x[..., 1, 1], x[..., 0, 1]
    y = torch.stack([0.5 * (a + b), 0.5 * (a - b), c], -1)
    return y / torch.sqrt(-ldot(y, y)).unsqueeze(-1)


def hyperboloid_to_spd2(y):
    a, b, c = y[..., 0] + y[..., 1], y[..., 0] - y[..., 1], y[..., 2]
    return torch.stack([torch.stack([a, c], -1), torch.stack([c, b], -1)], -2)


@pytest.mark.parametrize('seed', range(3))
@pytest.mark.parametrize('scale', [0.5, 1.0, 2.0])
def test_spd2_fixed_determinant_is_the_hyperbolic_plane(seed, scale):
    """test_isometry.py:52-83: SPD(2) matrices of determinant scale^2 are isometric to H^2 of curvature -1/2."""
    from graphembed import manifolds as M
    torch.manual_seed(seed)
    spd, lor = M.SymmetricPositiveDefinite(2), M.Lorentz(3)
    x = spd.rand(100, out=torch.empty(0, **F64), ir=1.0)
    x = x / x.det().sqrt().reshape(-1, 1, 1) * scale
    close(x.det(), torch.full((100, ), scale**2, **F64))
    close(x, spd.projx(x))
    y = unit_det_spd2_to_hyperboloid(x)
    close(spd.pdist(x), math.sqrt(2) * lor.pdist(y))


@pytest.mark.parametrize('seed', range(3))
@pytest.mark.parametrize('scale', [0.5, 1.0, 2.0])
def test_hyperboloid_to_spd2(seed, scale):
    """test_isometry.py:98-108."""
    from graphembed import manifolds as M
    torch.manual_seed(seed)
    spd, lor = M.SymmetricPositiveDefinite(2), M.Lorentz(3)
    x = lor.rand(100, out=torch.empty(0, **F64), ir=1.0) * scale
    y = hyperboloid_to_spd2(x)
    close(unit_det_spd2_to_hyperboloid(y), x / scale)
    close(spd.pdist(y), math.sqrt(2) * lor.pdist(x / scale))


@pytest.mark.parametrize('seed', range(3))
def test_sphere_and_hyperboloid_agree_near_the_pole(seed):
    """test_isometry.py:111-131: at distance ~1e-2 from the base point the two geometries coincide to 1e-4, through the
    reference's coordinate maps (x0 -> -1/x0, x_k -> -x_k/x0; back: x0 -> -1/x0, x_k -> x_k/x0)."""
    from graphembed import manifolds as M
    torch.manual_seed(seed)
    hyp, sph = M.Lorentz(3), M.Sphere(3)
    x = sph.rand(10, out=torch.empty(0, **F64), ir=1e-2)
    ds = sph.dist(sph.zero(10, out=torch.empty(0, **F64)), x)
    y = x.clone()                               # the sphere's base point is (-1, 0, 0): x0 ~ -1 lands on the sheet y0 > 0
    y[..., 1:] = -x[..., 1:] / x[..., :1]
    y[..., 0] = -1.0 / x[..., 0]
    close(ds, hyp.dist(hyp.zero(10, out=torch.empty(0, **F64)), y))
    x = hyp.rand(10, out=torch.empty(0, **F64), ir=1e-2)
    dh = hyp.dist(hyp.zero(10, out=torch.empty(0, **F64)), x)
    y = x.clone()
    y[..., 1:] = x[..., 1:] / x[..., :1]
    y[..., 0] = -1.0 / x[..., 0]
    close(sph.dist(sph.zero(10, out=torch.empty(0, **F64)), y), dh)


def test_euclidean_dim_and_batch_distances():
    """test_euclidean.py:11-24."""
    from graphembed import manifolds as M
    assert M.Euclidean(100).dim == 100 and M.Euclidean(10, 5, 2).dim == 100
    torch.manual_seed(0)
    x = torch.rand(20, 10, device='cuda')
    close(M.Euclidean(10).dist(x, x), np.zeros(20))


@pytest.mark.parametrize('n', [10, 100])
@pytest.mark.parametrize('d', range(10, 20))
def test_euclidean_pdists_scipy(n, d):
    """test_euclidean.py:27-33 (fp32, as the reference's default tensor type)."""
    from scipy.spatial.distance import pdist
    from graphembed import manifolds as M
    torch.manual_seed(n + d)
    x = torch.rand(n, d, device='cuda')
    close(M.Euclidean(d).pdist(x), pdist(x.cpu().numpy()))


def test_sphere_dist_poles():
    """test_sphere.py:9-15."""
    from graphembed import manifolds as M
    x = torch.zeros(5, device='cuda')
    y = torch.zeros(5, device='cuda')
    x[0], y[0] = 1.0, -1.0
    close(M.Sphere(5).dist(x, y), math.pi)


@pytest.mark.parametrize('d', [2, 3])
def test_symeig_of_the_identity_and_of_random_symmetric_matrices(d):
    """test_linalg.py:46-68: eigenvalues of I are ones; of uniform random symmetric matrices those of a LAPACK solve."""
    from graphembed import manifolds as M
    spd = M.SymmetricPositiveDefinite(d)
    eye = torch.eye(d, device='cuda').expand(10, -1, -1).contiguous()
    close(spd.symeig(eye), np.ones((10, d)))
    for n in range(10, 20):
        torch.manual_seed(n)
        x = torch.rand(n, d, d, device='cuda')
        x = 0.5 * (x + x.transpose(1, 2))
        close(spd.symeig(x), torch.linalg.eigvalsh(x.double().cpu()).float())
