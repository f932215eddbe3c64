"""GPU parity of the vector manifolds (Euclidean / Lorentz / Sphere), the product
embedding, the losses and the RSGD training loop against the golden vectors recorded
from the real reference."""
import itertools

import numpy as np
import pytest
import torch

from conftest import load_golden, sym

pytestmark = pytest.mark.gpu

DT = {'f32': torch.float32, 'f64': torch.float64}
KEYS = {'lorentz11': ('Lorentz', 11), 'lorentz6': ('Lorentz', 6), 'lorentz3': ('Lorentz', 3),
        'lorentz48': ('Lorentz', 48), 'sphere64': ('Sphere', 64), 'euclidean40': ('Euclidean', 40),
        'sphere6': ('Sphere', 6), 'euclidean10': ('Euclidean', 10)}
# Stated tolerances.  The distance maps are evaluated next to their singular points at the
# reference's own init (acosh at 1, acos at 1), where an fp32 ulp of the inner product moves
# d^2 by ~2e-7 absolute: hence absolute terms for fp32.  fp64 agrees to rounding.
ABS = {'f32': 1e-6, 'f64': 1e-12}
REL = {'f32': 2e-5, 'f64': 1e-10}
GREL = {'f32': 5e-4, 'f64': 1e-9}   # gradients, relative to max|grad| of the call


def make(key):
    from graphembed import manifolds as M
    name, n = KEYS[key]
    return getattr(M, name)(n)


def dev(a):
    return torch.from_numpy(np.array(a)).cuda()


def check_abs_rel(got, ref, dname, what, scale=1.0):
    got, ref = got.detach().double().cpu().numpy(), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    ok = np.isfinite(ref)
    bad = np.abs(got - ref)[ok] - scale * (ABS[dname] + REL[dname] * np.abs(ref[ok]))
    assert bad.max() <= 0, f'{what}: worst excess {bad.max():.3e}'


def check_rel(got, ref, tol, what):
    got = got.detach().double().cpu().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    ok = np.isfinite(ref)
    err = np.abs(got - ref)[ok].max() / max(np.abs(ref[ok]).max(), 1e-30)
    assert err <= tol, f'{what}: {err:.3e} > {tol:.1e}'


def exact_grads(key, x_np, g_np):
    """fp64 evaluation (oracle port) on exactly the given inputs: the yardstick that tells
    how far the REFERENCE's own fp32 result is from the truth at ill-conditioned points."""
    from oracle import ref_port as rp
    name, m = KEYS[key]
    port = rp.make(name.lower(), m)
    out = {}
    for sq in (True, False):
        xr = torch.from_numpy(np.array(x_np)).double().requires_grad_()
        d = port.pdist(xr, squared=sq)
        gr, = torch.autograd.grad((d * torch.from_numpy(np.array(g_np)).double()).sum(), xr)
        out[sq] = gr.numpy()
    return out


def check_grad(got, golden, exact, dname, what):
    """fp64: agree with the reference to rounding.  fp32: be at least as close to the exact
    gradient as the reference's own fp32 path is (x3 slack) — at the reference's init the
    maps are evaluated next to a singularity and BOTH fp32 results carry ~1e-3..1e-2 error."""
    if dname == 'f64':
        return check_rel(got, golden, GREL[dname], what)
    got = got.detach().double().cpu().numpy()
    scale = np.abs(exact).max()
    mine = np.abs(got - exact).max() / scale
    theirs = np.abs(np.asarray(golden, np.float64) - exact).max() / scale
    assert mine <= GREL[dname] + 3 * theirs, f'{what}: mine {mine:.3e} vs reference-fp32 {theirs:.3e}'


@pytest.mark.parametrize('key', list(KEYS))
@pytest.mark.parametrize('dname,init', list(itertools.product(DT, ['rand', 'wide'])))
def test_pdist_vs_reference_golden(key, dname, init):
    G = load_golden(key)
    man = make(key)
    for n in (33, 96):
        tag = f'{dname}/{init}/n{n}'
        if f'{tag}/x' not in G:
            continue
        g = dev(G[f'{tag}/g'])
        ex = exact_grads(key, G[f'{tag}/x'], G[f'{tag}/g']) if dname == 'f32' else None
        for gram in ([False, True] if man.use_gram else [False]):
            man.use_gram = gram
            x = dev(G[f'{tag}/x']).requires_grad_()
            d2 = man.pdist(x, squared=True)
            check_abs_rel(d2, G[f'{tag}/d2'], dname, f'd2 {tag} gram={gram}')
            gr, = torch.autograd.grad((d2 * g).sum(), x)
            check_grad(gr, G[f'{tag}/grad_d2'], ex and ex[True], dname, f'grad_d2 {tag}')
            d1 = man.pdist(x, squared=False)
            check_abs_rel(d1 * d1, np.asarray(G[f'{tag}/d1'], np.float64)**2, dname, f'd1^2 {tag} gram={gram}')
            gr, = torch.autograd.grad((d1 * g).sum(), x)
            check_grad(gr, G[f'{tag}/grad_d1'], ex and ex[False], dname, f'grad_d1 {tag}')
        dxy = man.dist(x.detach(), x.detach().flip(0), squared=True)
        check_abs_rel(dxy, G[f'{tag}/dist_xy'], dname, f'dist_xy {tag}')


@pytest.mark.parametrize('key', list(KEYS))
@pytest.mark.parametrize('dname,init', list(itertools.product(DT, ['rand', 'wide'])))
def test_maps_vs_reference_golden(key, dname, init):
    G = load_golden(key)
    man = make(key)
    tag = f'{dname}/{init}/n33'
    # (fp32 inner products of m terms: the rounding of both implementations grows with the dimension)
    tol = 2e-5 * max(1, KEYS[key][1] // 8) if dname == 'f32' else 1e-11
    x = dev(G[f'{tag}/x'])
    with torch.no_grad():
        check_rel(man.egrad2rgrad(x, dev(G[f'{tag}/grad_d2'])), G[f'{tag}/rgrad'], tol, 'egrad2rgrad')
        check_rel(man.norm(x, dev(G[f'{tag}/rgrad']), keepdim=True), G[f'{tag}/rgrad_norm'], tol, 'norm')
        u = dev(G[f'{tag}/u'])
        pu = man.proju(x, u)
        check_rel(pu, G[f'{tag}/proju'], tol, 'proju')
        check_rel(man.exp(x, pu), G[f'{tag}/exp'], tol, 'exp')
        check_rel(man.retr(x, pu), G[f'{tag}/retr'], tol, 'retr')
        # log at the reference init divides two O(1e-2) quantities that carry fp32 rounding
        check_rel(man.log(x, x.flip(0)), G[f'{tag}/log'], 5e-3 if dname == 'f32' else 1e-7, 'log')
        check_rel(man.projx(dev(G[f'{tag}/projx_in'])), G[f'{tag}/projx'], tol, 'projx')
        check_rel(man.transp(x, man.retr(x, pu), pu), G[f'{tag}/transp'], tol * 5, 'transp')


@pytest.mark.parametrize('key', list(KEYS))
@pytest.mark.parametrize('dname', list(DT))
def test_rsgd_vs_reference_golden(key, dname):
    from graphembed.modules import ManifoldParameter
    from graphembed.optim import RiemannianSGD
    G = load_golden(key)
    man = make(key)
    base = f'{dname}/rsgd'
    tol = 5e-5 * max(1, KEYS[key][1] // 8) if dname == 'f32' else 1e-10
    for exact, clip, mom in itertools.product([0, 1], [0, 1], [0, 1]):
        tag = f'{base}/exact{exact}_clip{clip}_mom{mom}'
        p = ManifoldParameter(dev(G[f'{base}/x0']), manifold=man)
        opt = RiemannianSGD([p], lr=0.05, momentum=0.9 if mom else 0, dampening=0.1 if mom else 0,
                            max_grad_norm=2.0 if clip else None, exact=bool(exact))
        p.grad = dev(G[f'{base}/g1'])
        opt.step()
        check_rel(p.data, G[f'{tag}/x1'], tol, tag + '/x1')
        p.grad = dev(G[f'{base}/g2'])
        opt.step()
        check_rel(p.data, G[f'{tag}/x2'], tol, tag + '/x2')
        if mom:
            check_rel(opt.state[p]['momentum_buffer'], G[f'{tag}/buf2'], tol, tag + '/buf2')


@pytest.mark.parametrize('key,n', [('lorentz11', 1000), ('sphere6', 700), ('euclidean10', 515), ('lorentz3', 300)])
@pytest.mark.parametrize('dname', list(DT))
def test_pdist_vs_oracle_seeded(key, n, dname):
    from oracle import ref_port as rp
    name, m = KEYS[key]
    port = rp.make(name.lower(), m)
    man = make(key)
    gen = torch.Generator().manual_seed(n)
    x64 = port.rand(n, ir=0.3 if name != 'Sphere' else 0.5, dtype=torch.float64, generator=gen)
    g64 = torch.randn(n * (n - 1) // 2, dtype=torch.float64, generator=gen)
    xr = x64.clone().requires_grad_()
    ref = port.pdist(xr, squared=True)
    ref_g, = torch.autograd.grad((ref * g64).sum(), xr)
    for gram in ([False, True] if man.use_gram else [False]):
        man.use_gram = gram
        x = x64.to(DT[dname]).cuda().requires_grad_()
        d2 = man.pdist(x, squared=True)
        check_abs_rel(d2, ref.detach().numpy(), dname, f'd2 {key} gram={gram}')
        gr, = torch.autograd.grad((d2 * g64.to(DT[dname]).cuda()).sum(), x)
        check_rel(gr, ref_g.numpy(), GREL[dname], f'grad {key}')


@pytest.mark.parametrize('key', ['lorentz11', 'sphere6', 'euclidean10'])
def test_row_sharding(key):
    from graphembed import _backend as B
    man = make(key)
    torch.manual_seed(5)
    n = 611
    x = man.rand(n, out=torch.empty(0, device='cuda'), ir=0.3)
    g = torch.randn(n * (n - 1) // 2, device='cuda')
    xr = x.clone().requires_grad_()
    full = man.pdist(xr, squared=True)
    gfull, = torch.autograd.grad((full * g).sum(), xr)
    for world in (2, 5):
        parts, gsum = [], torch.zeros_like(x)
        for r in range(world):
            rb, re = B.shard_rows(n, world, r)
            xr = x.clone().requires_grad_()
            part = man.pdist(xr, squared=True, rows=(rb, re))
            lo, hi = B.pair_offset(n, rb), B.pair_offset(n, re)
            gp, = torch.autograd.grad((part * g[lo:hi]).sum(), xr)
            parts.append(part.detach())
            gsum += gp
        assert torch.equal(torch.cat(parts), full.detach())
        check_rel(gsum, gfull.cpu().numpy(), 1e-5, 'sum of shard grads')


@pytest.mark.parametrize('dname', list(DT))
def test_product_embedding_and_losses(dname):
    """ManifoldEmbedding.compute_dists over H^5 x S^5 x SPD(2) (config 4's product) incl. the
    gradients of the learnable scales, full batch and node mini-batch (modules.py:84-88)."""
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import QuotientLoss, StressLoss
    G = load_golden('callers')
    torch.set_default_dtype(DT[dname])
    try:
        with torch.device('cuda'):
            emb = ManifoldEmbedding(33, [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)])
    finally:
        torch.set_default_dtype(torch.float32)
    idx = dev(G[f'{dname}/product/idx'])
    for tag, ii in [('full', None), ('batch', idx)]:
        base = f'{dname}/product/{tag}'
        with torch.no_grad():
            for k in range(3):
                emb.xs[k].copy_(dev(G[f'{base}/x_{k}']))
                emb.scales[k].fill_(float(G[f'{base}/scales'][k]))
        md = emb.compute_dists(ii)
        # the SPD(2) factor of the reference carries its +1e-8 Cholesky fudge (fast.py:103)
        check_abs_rel(md, G[f'{base}/d2'], dname, f'product d2 {tag}', scale=3 if dname == 'f32' else 1e5)
        grads = torch.autograd.grad((md * dev(G[f'{base}/g'])).sum(), list(emb.xs) + list(emb.scales))
        for k in range(3):
            ref = G[f'{base}/grad_x_{k}']
            check_rel(grads[k], sym(ref) if k == 2 else ref, max(GREL[dname], 5e-6) if k == 2 else GREL[dname],
                      f'product grad_x_{k} {tag}')
            check_rel(grads[3 + k], G[f'{base}/grad_s_{k}'], max(GREL[dname], 5e-6) if k == 2 else GREL[dname],
                      f'product grad_s_{k} {tag}')
    gd, md = dev(G[f'{dname}/loss/gd']), dev(G[f'{dname}/loss/md'])
    check_rel(StressLoss()(gd, md), G[f'{dname}/loss/stress'], 1e-5, 'stress')
    check_rel(QuotientLoss()(gd, md, epoch=3, alpha=1.7), G[f'{dname}/loss/quotient'], 1e-5, 'quotient')


@pytest.mark.parametrize('case', ['euclidean10', 'lorentz11', 'spd3', 'product'])
@pytest.mark.parametrize('loss_name', ['stress', 'quotient'])
def test_tree40_training_trace(case, loss_name):
    """config[0]-style plumbing: 20 full-batch epochs on tree40 in fp64 with the reference's
    production optimizer settings (experiments/run_grid.py:24-36), through this package's
    ManifoldEmbedding + RiemannianSGD + losses, against the reference's loss trace."""
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import QuotientLoss, StressLoss
    from graphembed.optim import RiemannianSGD
    G = load_golden('callers')
    mk = {'euclidean10': lambda: [M.Euclidean(10)], 'lorentz11': lambda: [M.Lorentz(11)],
          'spd3': lambda: [M.SymmetricPositiveDefinite(3)],
          'product': lambda: [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)]}[case]
    base = f'tree40/{case}/{loss_name}'
    torch.set_default_dtype(torch.float64)
    try:
        with torch.device('cuda'):
            emb = ManifoldEmbedding(40, mk())
        with torch.no_grad():
            for k, x in enumerate(emb.xs):
                x.copy_(dev(G[f'{base}/x0_{k}']))
        target = dev(G['tree40/target'])
        opt = RiemannianSGD(list(emb.xs), lr=0.01, exact=True, max_grad_norm=20)
        opt_s = RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)
        fn = StressLoss() if loss_name == 'stress' else QuotientLoss()
        losses = []
        for epoch in range(20):
            loss = fn(target, emb.compute_dists(None), epoch=epoch, alpha=1.0)
            opt.zero_grad()
            opt_s.zero_grad()
            loss.backward()
            opt.step()
            opt_s.step()
            losses.append(loss.item())
    finally:
        torch.set_default_dtype(torch.float32)
    # SPD factors differ from the reference by its eps-fudged closed forms (~1e-7)
    tol = 1e-7 if case in ('euclidean10', 'lorentz11') else 2e-5
    check_rel(np.array(losses), G[f'{base}/losses'], tol, 'loss trace')
    for k, x in enumerate(emb.xs):
        check_rel(x.data, G[f'{base}/x20_{k}'], max(tol, 1e-8) * 10, f'x20_{k}')
    check_rel(np.array([s.item() for s in emb.scales]), G[f'{base}/scales20'], 1e-6, 'scales')


def test_isometry_spd2_lorentz3():
    """tests/test_isometry.py:51-63 of the reference: SPD(2)/det=1 is isometric to H^2 with
    curvature -1/2: spd.pdist(x) == sqrt(2) * lorentz.pdist(phi(x))."""
    from graphembed import manifolds as M
    torch.manual_seed(1)
    n = 200
    h = M.Lorentz(3).rand(n, out=torch.empty(0, dtype=torch.float64, device='cuda'), ir=1.0)
    t, a, b = h[:, 0], h[:, 1], h[:, 2]
    x = torch.stack([torch.stack([t + a, b], -1), torch.stack([b, t - a], -1)], -2)  # det = t^2-a^2-b^2 = 1
    ds = M.SymmetricPositiveDefinite(2).pdist(x)
    dl = M.Lorentz(3).pdist(h)
    np.testing.assert_allclose(ds.cpu().numpy(), (2 ** 0.5) * dl.cpu().numpy(), rtol=1e-7, atol=1e-9)


def test_sphere_antipodal_and_euclid_scipy():
    """tests/test_sphere.py:9-15 and tests/test_euclidean.py:27-33 of the reference."""
    from scipy.spatial.distance import pdist as sp_pdist
    from graphembed import manifolds as M
    x = torch.randn(10, 6, dtype=torch.float64, device='cuda')
    x = x / x.norm(dim=-1, keepdim=True)
    d = M.Sphere(6).dist(x, -x)
    np.testing.assert_allclose(d.cpu().numpy(), np.pi, atol=1e-4)
    for n, d_ in [(10, 10), (100, 19)]:
        x = torch.randn(n, d_, dtype=torch.float64, device='cuda')
        np.testing.assert_allclose(M.Euclidean(d_).pdist(x).cpu().numpy(), sp_pdist(x.cpu().numpy()), atol=1e-10)


def test_optim_sphere_dominant_eigenvector():
    """tests/test_optim.py:13-41 of the reference: RSGD on the sphere finds the dominant
    eigenvector; the iterate stays on the manifold."""
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldParameter
    from graphembed.optim import RiemannianSGD
    torch.manual_seed(0)
    n = 8
    a = torch.randn(n, n, dtype=torch.float64, device='cuda')
    a = a @ a.T
    man = M.Sphere(n)
    x = ManifoldParameter(man.rand_uniform(1, out=a)[0], manifold=man)
    opt = RiemannianSGD([x], lr=1e-2)
    for _ in range(2000):
        opt.zero_grad()
        loss = -(x @ a @ x)
        loss.backward()
        opt.step()
        assert abs(x.detach().norm().item() - 1) < 1e-6
    w, v = torch.linalg.eigh(a)
    assert abs(abs((v[:, -1] @ x.detach()).item()) - 1) < 1e-3


@pytest.mark.parametrize('n', [3, 4, 5])
@pytest.mark.parametrize('optim', ['rsgd', 'radam'])
@pytest.mark.parametrize('seed', [0, 1])
def test_reference_optim_test_dominant_eigenvector(n, optim, seed):
    """graphembed/tests/test_optim.py:13-41 with its own parametrisation, sizes, learning rate, iteration count and
    tolerances (atol 1e-4), for RiemannianSGD AND RiemannianAdam: 200 steps on the sphere find the dominant eigenvector of
    a random symmetric matrix; the iterate has unit norm after every step."""
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldParameter
    from graphembed.optim import RiemannianAdam, RiemannianSGD
    torch.manual_seed(seed)
    a = torch.rand(n, n, dtype=torch.float64, device='cuda')
    a = 0.5 * (a + a.T)                                           # conftest.py rand_sym
    man = M.Sphere(n)
    x = ManifoldParameter(man.rand(1, out=torch.empty(0, dtype=torch.float64, device='cuda'))[0], manifold=man)
    opt = RiemannianSGD([x], lr=1e-1) if optim == 'rsgd' else RiemannianAdam([x], lr=1e-1)
    for _ in range(200):
        opt.zero_grad()
        loss = -torch.einsum('i,ij,j', x, a, x)
        loss.backward()
        opt.step()
        assert abs(1.0 - x.detach().norm().item()) <= 1e-4
    x_opt = x.detach()
    w, v = torch.linalg.eigh(a)
    assert ((x_opt / v[:, -1]).abs() - 1).abs().max().item() <= 1e-4
    assert abs((a @ x_opt).norm().item() - w[-1].item()) <= 1e-4


@pytest.mark.parametrize('case', ['euclidean10', 'lorentz11', 'sphere6', 'lorentz3'])
@pytest.mark.parametrize('dname', ['f32', 'f64'])
@pytest.mark.parametrize('loss_name', ['stress', 'quotient'])
def test_fused_loss_equals_unfused_path(case, dname, loss_name):
    """mm_vec_pdist_loss (one pass: loss, d/dx, d/dscale) vs compute_dists -> objective -> backward
    of this library (itself checked against the reference above); row shards sum to the whole."""
    from graphembed import _backend as B
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import QuotientLoss, StressLoss
    mk = {'euclidean10': lambda: M.Euclidean(10), 'lorentz11': lambda: M.Lorentz(11),
          'sphere6': lambda: M.Sphere(6), 'lorentz3': lambda: M.Lorentz(3)}[case]
    dt = {'f32': torch.float32, 'f64': torch.float64}[dname]
    n = 700
    torch.manual_seed(3)
    torch.set_default_dtype(dt)
    try:
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, [mk()])
            with torch.no_grad():  # spread the points: at the reference init every distance is ~0.1
                emb.perturb(0.5)
    finally:
        torch.set_default_dtype(torch.float32)
    fn, kw = (StressLoss(), {}) if loss_name == 'stress' else (QuotientLoss(), dict(epoch=2, alpha=1.3))
    md = emb.compute_dists(None).detach()
    gen = torch.Generator(device='cuda').manual_seed(1)
    target = md * (0.5 + torch.rand(md.shape, dtype=dt, device='cuda', generator=gen))
    if kw:  # keep away from the |.| kinks
        for _ in range(8):
            ag = target * kw['alpha']
            near = ((md / ag - 1).abs() < 0.05) | ((ag / (md + 1 / (kw['epoch'] + 1)) - 1).abs() < 0.05)
            target = torch.where(near, target * 1.25, target)
        assert not near.any()
    ref = fn(target, emb.compute_dists(None), **kw)
    rgx, rgs = torch.autograd.grad(ref, [emb.xs[0], emb.scales[0]])
    loss = emb.fused_objective(fn, target, None, **kw)
    assert loss is not None
    gx, gs = torch.autograd.grad(loss * 3.0, [emb.xs[0], emb.scales[0]])
    tol = 5e-5 if dname == 'f32' else 1e-10
    assert abs(loss.item() - ref.item()) <= tol * abs(ref.item())
    check_rel(gx, 3 * rgx.double().cpu().numpy(), tol * 4, 'fused vs unfused grad_x')
    assert abs(gs.item() - 3 * rgs.item()) <= (2e-3 if dname == 'f32' else 1e-9) * abs(3 * rgs.item())
    tot, gsum, ssum = 0.0, torch.zeros_like(gx), 0.0
    for r in range(3):
        rows = B.shard_rows(n, 3, r)
        lo, hi = B.pair_offset(n, rows[0]), B.pair_offset(n, rows[1])
        part = emb.fused_objective(fn, target[lo:hi], None, rows=rows, **kw)
        pgx, pgs = torch.autograd.grad(part, [emb.xs[0], emb.scales[0]])
        tot, gsum, ssum = tot + part.item(), gsum + pgx, ssum + pgs.item()
    assert abs(tot - ref.item()) <= tol * abs(ref.item())
    check_rel(gsum, rgx.double().cpu().numpy(), tol * 4, 'sum of fused shard grads')


@pytest.mark.parametrize('case', ['euclidean10', 'lorentz11', 'product', 'product_graph'])
@pytest.mark.parametrize('loss_name', ['stress', 'quotient'])
def test_tree40_training_trace_fused(case, loss_name):
    """The reference's 20-epoch tree40 loss traces (golden, recorded from the real reference) driven through
    the fused kernels: single factors (mm_vec_pdist_loss), the H^5 x S^5 x SPD(2) product through the
    mixed-manifold pair kernel + the multi-parameter RSGD launch — and, `product_graph`, the same as
    ONE captured HIP graph replayed for 19 epochs with the quotient schedule in device memory."""
    from graphembed import manifolds as M
    from graphembed import unit_seed
    from graphembed.graphed import GraphedTrainStep
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import QuotientLoss, StressLoss
    from graphembed.optim import RiemannianSGD
    G = load_golden('callers')
    graphed = case == 'product_graph'
    gcase = 'product' if graphed else case
    mk = {'euclidean10': lambda: [M.Euclidean(10)], 'lorentz11': lambda: [M.Lorentz(11)],
          'product': lambda: [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)]}[gcase]
    base = f'tree40/{gcase}/{loss_name}'
    torch.set_default_dtype(torch.float64)
    try:
        with torch.device('cuda'):
            emb = ManifoldEmbedding(40, mk())
        with torch.no_grad():
            for k, x in enumerate(emb.xs):
                x.copy_(dev(G[f'{base}/x0_{k}']))
        target = dev(G['tree40/target'])
        opt = RiemannianSGD(list(emb.xs), lr=0.01, exact=True, max_grad_norm=20)
        opt_s = RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)
        fn = StressLoss() if loss_name == 'stress' else QuotientLoss()
        losses = []
        if graphed:
            if loss_name == 'quotient':
                fn.on_device('cuda')
                fn.set_epoch(0, 1.0)
            step = GraphedTrainStep(lambda: emb.fused_objective(fn, target, None, epoch=0, alpha=1.0),
                                    [opt, opt_s], warmup=1).capture()
            losses.append(step.warmup_losses[0].item())
            for epoch in range(1, 20):
                if loss_name == 'quotient':
                    fn.set_epoch(epoch, 1.0)
                losses.append(step().item())
        else:
            for epoch in range(20):
                loss = emb.fused_objective(fn, target, None, epoch=epoch, alpha=1.0)
                assert loss is not None
                opt.zero_grad()
                opt_s.zero_grad()
                loss.backward(unit_seed(loss) if epoch % 2 else None)   # both seeds
                opt.step()
                opt_s.step()
                losses.append(loss.item())
    finally:
        torch.set_default_dtype(torch.float32)
    # SPD factors differ from the reference by its eps-fudged closed forms (~1e-7)
    tol = 1e-7 if gcase in ('euclidean10', 'lorentz11') else 2e-5
    check_rel(np.array(losses), G[f'{base}/losses'], tol, 'loss trace')
    for k, x in enumerate(emb.xs):
        check_rel(x.data, G[f'{base}/x20_{k}'], max(tol, 1e-8) * 10, f'x20_{k}')
    check_rel(np.array([s.item() for s in emb.scales]), G[f'{base}/scales20'], 1e-6, 'scales')


@pytest.mark.parametrize('dname', ['f32', 'f64'])
@pytest.mark.parametrize('exact,clip', [(False, None), (True, 0.05)])
def test_rsgd_multi_parameter_launch_equals_single_steps(dname, exact, clip):
    """RiemannianSGD steps all vector-space parameters of a momentum-free group with one launch
    (mm_vec_rsgd_step_multi): bit-identical to the one-kernel-per-parameter updates (rsgd.py:52-82)."""
    from graphembed import _backend as B
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldParameter
    from graphembed.optim import RiemannianSGD
    from graphembed.optim._common import FLAT
    dt = {'f32': torch.float32, 'f64': torch.float64}[dname]
    torch.manual_seed(11)
    mans = [M.Euclidean(5), M.Lorentz(6), M.Sphere(4), M.SymmetricPositiveDefinite(2)]
    pts = [man.rand(37 + 5 * k, out=torch.empty(0, dtype=dt, device='cuda')) for k, man in enumerate(mans)]
    grads = [torch.randn_like(x) for x in pts]
    scal = [torch.randn((), dtype=dt, device='cuda') for _ in range(3)] + [torch.randn(4, 3, dtype=dt, device='cuda')]
    sgrads = [torch.randn_like(s) for s in scal]
    want = [man.rsgd_step(x.clone(), g, lr=0.1, max_grad_norm=clip, exact=exact) for man, x, g in zip(mans, pts, grads)]
    want += [FLAT.rsgd_step(s.clone(), g, lr=0.1, max_grad_norm=clip, exact=exact) for s, g in zip(scal, sgrads)]
    params = [ManifoldParameter(x.clone(), manifold=man) for man, x in zip(mans, pts)]
    params += [torch.nn.Parameter(s.clone()) for s in scal]
    for p, g in zip(params, grads + sgrads):
        p.grad = g.clone()
    lib, calls = B.lib(), []
    orig = lib.call
    lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        RiemannianSGD(params, lr=0.1, exact=exact, max_grad_norm=clip).step()
    finally:
        del lib.call
    assert calls.count('mm_vec_rsgd_step_multi') == 1 and calls.count('mm_vec_rsgd_step') == 0, calls
    assert calls.count('mm_spd_rsgd_step') == 1
    for p, w in zip(params, want):
        assert torch.equal(p.detach(), w)


def test_rsgd_momentum_uses_fused_kernels():
    """momentum > 0: one launch per parameter (mm_*_rsgd_momentum_step), no per-map kernels; the golden traces
    of test_rsgd_vs_reference_golden / test_spd_gpu pin the values."""
    from graphembed import _backend as B
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldParameter
    from graphembed.optim import RiemannianSGD
    torch.manual_seed(2)
    mans = [M.Lorentz(6), M.Sphere(5), M.SymmetricPositiveDefinite(3)]
    ps = [ManifoldParameter(man.rand(30, out=torch.empty(0, device='cuda')), manifold=man) for man in mans]
    ps.append(torch.nn.Parameter(torch.randn(3, device='cuda')))
    opt = RiemannianSGD(ps, lr=0.05, momentum=0.9, dampening=0.1, max_grad_norm=2.0, exact=True)
    lib, calls = B.lib(), []
    orig = lib.call
    lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        for _ in range(2):
            for p in ps:
                g = torch.randn_like(p)
                p.grad = g + g.transpose(-2, -1) if p.ndim == 3 else g
            opt.step()
    finally:
        del lib.call
    assert calls.count('mm_vec_rsgd_momentum_step') == 6 and calls.count('mm_spd_rsgd_momentum_step') == 2, calls
    assert len(calls) == 8
    for p in ps:
        assert bool(torch.isfinite(p).all()) and bool(torch.isfinite(opt.state[p]['momentum_buffer']).all())


def test_gram_default_falls_back_outside_its_range():
    """Lorentz / Sphere route pdist through the matrix-core Gram kernels by default; dimensions those do not
    serve (fp64 beyond m = 16) take the VALU kernels instead of failing (found by tests/fuzz_pdist.py)."""
    from graphembed import manifolds as M
    from oracle import exact
    torch.manual_seed(0)
    for man, kind in ((M.Lorentz(20), 'lorentz'), (M.Sphere(24), 'sphere')):
        x = man.rand(50, out=torch.empty(0, dtype=torch.float64, device='cuda'), ir=0.3) if kind == 'lorentz' \
            else torch.nn.functional.normalize(torch.randn(50, 24, dtype=torch.float64, device='cuda'), dim=-1)
        x.requires_grad_()
        g = torch.randn(50 * 49 // 2, dtype=torch.float64, device='cuda')
        d2 = man.pdist(x, squared=True)
        gr, = torch.autograd.grad(d2, x, g)
        ref = exact.vec_pdist(kind, x.detach().cpu().numpy())
        rg = exact.vec_pdist_grad(kind, x.detach().cpu().numpy(), g.cpu().numpy())
        assert np.abs(d2.detach().cpu().numpy() - ref).max() <= 1e-11
        assert np.abs(gr.cpu().numpy() - rg).max() <= 1e-8 * np.abs(rg).max()
