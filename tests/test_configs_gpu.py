"""Full-size parity on the BASELINE.json configurations, GPU (through the C ABI) vs the plain-C
fp64 checker oracle/exact.c on the same seeded synthetic inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def spd_points(n, d, seed, init='rand'):
    from oracle import ref_port as rp
    gen = torch.Generator().manual_seed(seed)
    if init == 'rand':
        return rp.SPD(d).rand(n, dtype=torch.float64, generator=gen)
    a = torch.rand(n, d, d, dtype=torch.float64, generator=gen)
    return a @ a.transpose(1, 2) + torch.eye(d, dtype=torch.float64)


def check(d2, grad, ref_d2, ref_grad, dtol, gtol, what):
    d2 = d2.double().cpu().numpy()
    bad = np.abs(d2 - ref_d2) - (dtol[0] + dtol[1] * np.abs(ref_d2))
    assert bad.max() <= 0, f'{what}: d2 worst excess {bad.max():.3e}'
    g = grad.double().cpu().numpy()
    err = np.abs(g - ref_grad).max() / np.abs(ref_grad).max()
    assert err <= gtol, f'{what}: grad rel err {err:.3e} > {gtol:.1e}'


@pytest.mark.parametrize('n,d,init,dtype', [
    (5000, 3, 'rand', torch.float32),     # config 3 / the BASELINE metric: grqc-class graph -> SPD(3)
    (5000, 3, 'wide', torch.float32),
    (4158, 3, 'rand', torch.float64),     # grqc's actual node count, run.py's fp64 default
    (2274, 4, 'rand', torch.float32),     # config 5: bio-wormnet actual size -> SPD(4)
    (1025, 2, 'wide', torch.float32),     # csphd-size SPD(2) factor
])
def test_spd_full_size(n, d, init, dtype):
    from graphembed.manifolds import SymmetricPositiveDefinite as SPD
    from oracle import exact
    x64 = spd_points(n, d, 42, init)
    g64 = torch.randn(n * (n - 1) // 2, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    xin = x64.to(dtype)                      # the checker sees exactly the (rounded) inputs of the kernel
    ref_d2 = exact.spd_pdist(xin.double().numpy())
    ref_g = exact.spd_pdist_grad(xin.double().numpy(), g64.to(dtype).double().numpy())
    x = xin.cuda().requires_grad_()
    d2 = SPD(d).pdist(x, squared=True)
    gr, = torch.autograd.grad(d2, x, g64.to(dtype).cuda())
    f32 = dtype == torch.float32
    check(d2.detach(), gr, ref_d2, ref_g, (1e-6, 2e-5) if f32 else (1e-13, 1e-11), 2e-5 if f32 else 1e-10,
          f'SPD({d}) n={n} {init} {dtype}')


@pytest.mark.parametrize('kind,n,m,dtype', [
    ('euclidean', 40, 10, torch.float64),     # config 1: tree40 -> R^10
    ('lorentz', 4039, 11, torch.float32),     # config 2: facebook -> H^10
    ('lorentz', 4039, 11, torch.float64),
    ('sphere', 1025, 6, torch.float32),       # config 4 factors (csphd): H^5 x S^5 x SPD(2)
    ('lorentz', 1025, 6, torch.float32),
])
def test_vec_full_size(kind, n, m, dtype):
    from graphembed import manifolds as M
    from oracle import exact
    from oracle import ref_port as rp
    gen = torch.Generator().manual_seed(7)
    x64 = rp.make(kind, m).rand(n, ir=0.3, dtype=torch.float64, generator=gen)
    g64 = torch.randn(n * (n - 1) // 2, dtype=torch.float64, generator=gen)
    xin = x64.to(dtype)
    ref_d2 = exact.vec_pdist(kind, xin.double().numpy())
    ref_g = exact.vec_pdist_grad(kind, xin.double().numpy(), g64.to(dtype).double().numpy())
    man = {'euclidean': M.Euclidean, 'lorentz': M.Lorentz, 'sphere': M.Sphere}[kind](m)
    x = xin.cuda().requires_grad_()
    d2 = man.pdist(x, squared=True)
    gr, = torch.autograd.grad(d2, x, g64.to(dtype).cuda())
    f32 = dtype == torch.float32
    check(d2.detach(), gr, ref_d2, ref_g, (1e-6, 2e-5) if f32 else (1e-13, 1e-11), 2e-4 if f32 else 1e-10,
          f'{kind}({m}) n={n} {dtype}')


def test_spd4_stress_size_properties():
    """config 5 at its nominal stress size (n = 16 384, 134 M pairs, SPD(4)): no oracle that large —
    size-independent properties instead: finiteness, the Euler identity of the scale-invariant
    distance (sum_i <grad_i, X_i> = 0), and agreement of sampled pairs with the element-wise kernel."""
    from graphembed.manifolds import SymmetricPositiveDefinite as SPD
    n, d = 16384, 4
    man = SPD(d)
    torch.manual_seed(0)
    x = man.rand(n, out=torch.empty(0, device='cuda')).requires_grad_()
    P = n * (n - 1) // 2
    d2 = man.pdist(x, squared=True)
    assert d2.numel() == P and bool(torch.isfinite(d2).all())
    g = torch.randn(P, device='cuda')
    gr, = torch.autograd.grad(d2, x, g)
    assert bool(torch.isfinite(gr).all())
    euler = (gr.double() * x.detach().double()).sum().abs().item()
    assert euler <= 1e-4 * gr.double().abs().sum().item()
    idx = torch.randint(0, n, (2, 4000), device='cuda')
    i, j = idx.min(0).values, idx.max(0).values
    keep = i < j
    i, j = i[keep], j[keep]
    k = i * (2 * n - i - 1) // 2 + (j - i - 1)
    # (the number of Jacobi sweeps is decided per wavefront, so the two kernels may differ by rounding)
    ref = man.dist(x.detach()[i], x.detach()[j], squared=True)
    assert (d2.detach()[k] - ref).abs().max().item() <= 1e-7 + 1e-5 * ref.max().item()


@pytest.mark.parametrize('case', ['spd3_fused', 'product'])
def test_graphed_train_step_matches_eager(case):
    """A training step captured as one HIP graph (graphembed.graphed) advances the parameters
    exactly like the eager loop (warm-up steps are ordinary steps; recording a step does not execute it)."""
    from graphembed import manifolds as M
    from graphembed.graphed import GraphedTrainStep
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import StressLoss
    from graphembed.optim import RiemannianAdam, RiemannianSGD
    n = 200
    mans = {'spd3_fused': lambda: [M.SymmetricPositiveDefinite(3)],
            'product': lambda: [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)]}[case]
    torch.set_default_dtype(torch.float64)
    try:
        def build():
            torch.manual_seed(5)
            with torch.device('cuda'):
                emb = ManifoldEmbedding(n, mans())
            opts = [RiemannianSGD(list(emb.xs), lr=0.01, exact=True, max_grad_norm=20),
                    RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)]
            return emb, opts
        torch.manual_seed(1)
        target = torch.rand(n * (n - 1) // 2, device='cuda') * 0.9 + 0.1
        fn = StressLoss()

        def loss_of(emb):
            if case == 'spd3_fused':
                return emb.fused_objective(fn, target, None)
            return fn(target, emb.compute_dists(None))
        emb_e, opts_e = build()
        losses_e = []
        for _ in range(6):
            for o in opts_e:
                o.zero_grad()
            loss = loss_of(emb_e)
            loss.backward()
            for o in opts_e:
                o.step()
            losses_e.append(loss.item())
        emb_g, opts_g = build()
        step = GraphedTrainStep(lambda: loss_of(emb_g), opts_g, warmup=2).capture()  # 2 eager steps, then record
        after_capture = [x.detach().clone() for x in emb_g.xs]
        torch.cuda.synchronize()
        for x, x_ in zip(emb_g.xs, after_capture):
            assert torch.equal(x.detach(), x_), 'recording a step must not execute it'
        losses_g = [l.item() for l in step.warmup_losses] + [step().item() for _ in range(4)]
        np.testing.assert_allclose(losses_g, losses_e, rtol=1e-9)
        for a, b in zip(emb_g.xs, emb_e.xs):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-8, atol=1e-10)
        for a, b in zip(emb_g.scales, emb_e.scales):
            assert abs(a.item() - b.item()) <= 1e-10
        # RiemannianAdam keeps its step counter in device memory: graph replay == eager as well
        emb_a, _ = build()
        emb_b, _ = build()
        opt_a = [RiemannianAdam(list(emb_a.xs), lr=1e-3, max_grad_norm=20)]
        opt_b = [RiemannianAdam(list(emb_b.xs), lr=1e-3, max_grad_norm=20)]
        for _ in range(5):
            opt_a[0].zero_grad()
            la = loss_of(emb_a)
            la.backward()
            opt_a[0].step()
        step_b = GraphedTrainStep(lambda: loss_of(emb_b), opt_b, warmup=2).capture()
        lb = [step_b().item() for _ in range(3)][-1]
        assert abs(lb - la.item()) <= 1e-9 * abs(la.item())
        for a, b in zip(emb_a.xs, emb_b.xs):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-8, atol=1e-10)

        class _HostState(torch.optim.SGD):
            pass
        with pytest.raises(TypeError):
            GraphedTrainStep(lambda: loss_of(emb_g), [_HostState(list(emb_g.scales), lr=1e-3)])
    finally:
        torch.set_default_dtype(torch.float32)


@pytest.mark.parametrize('dname', ['f32', 'f64'])
@pytest.mark.parametrize('loss_name', ['stress', 'quotient'])
def test_product_fused_objective_equals_unfused(dname, loss_name):
    """csphd-style product (Lorentz x Sphere x SPD(2)), other mixes and a Grassmann single factor: the
    fused objective — the single mixed-manifold pair kernel (mm_product_pairs_loss) where it applies,
    else per-factor pair kernels around ONE loss kernel (mm_product_loss) — gives the loss and all
    gradients of compute_dists -> objective -> backward; row shards sum to the whole."""
    from graphembed import _backend as B
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import QuotientLoss, StressLoss
    dt = {'f32': torch.float32, 'f64': torch.float64}[dname]
    n = 300
    fn, kw = (StressLoss(), {}) if loss_name == 'stress' else (QuotientLoss(), dict(epoch=1, alpha=1.2))
    tol = 1e-4 if dname == 'f32' else 1e-10
    from graphembed.modules import _pair_kernel_factors
    cases = [([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], True),
             ([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], False),
             ([M.Euclidean(5), M.SymmetricPositiveDefinite(3)], True),
             ([M.Lorentz(3), M.Euclidean(2), M.Sphere(16)], True),
             ([M.Sphere(3), M.Lorentz(16), M.Euclidean(7), M.SymmetricPositiveDefinite(3)], True),  # largest launch
             ([M.SymmetricPositiveDefinite(2), M.SymmetricPositiveDefinite(3)], True),  # two SPD: per-factor path
             ([M.Grassmann(5, 2)], True)]
    assert _pair_kernel_factors(cases[0][0]) is not None and _pair_kernel_factors(cases[5][0]) is None
    for mans, pair_kernel in cases:
        torch.manual_seed(4)
        torch.set_default_dtype(dt)
        try:
            with torch.device('cuda'):
                emb = ManifoldEmbedding(n, mans)
                with torch.no_grad():
                    emb.perturb(0.3)
        finally:
            torch.set_default_dtype(torch.float32)
        emb.pair_kernel = pair_kernel
        params = list(emb.xs) + list(emb.scales)
        md = emb.compute_dists(None).detach()
        gen = torch.Generator(device='cuda').manual_seed(2)
        target = md * (0.5 + torch.rand(md.shape, dtype=dt, device='cuda', generator=gen))
        if kw:
            for _ in range(8):
                ag = target * kw['alpha']
                near = ((md / ag - 1).abs() < 0.05) | ((ag / (md + 1 / (kw['epoch'] + 1)) - 1).abs() < 0.05)
                target = torch.where(near, target * 1.25, target)
            assert not near.any()
        ref = fn(target, emb.compute_dists(None), **kw)
        rg = torch.autograd.grad(ref, params)
        lib, calls = B.lib(), []
        orig = lib.call
        lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            loss = emb.fused_objective(fn, target, None, **kw)
        finally:
            del lib.call
        assert loss is not None
        assert ('mm_product_pairs_loss' in calls) == (pair_kernel and _pair_kernel_factors(mans) is not None
                                                      and len(mans) > 1), calls
        g = torch.autograd.grad(loss * 2.0, params)
        assert abs(loss.item() - ref.item()) <= tol * abs(ref.item())
        for a, b in zip(g, rg):
            err = (a - 2 * b).abs().max().item() / max(b.abs().max().item() * 2, 1e-30)
            assert err <= 20 * tol, err
        tot, gsum = 0.0, [torch.zeros_like(p) for p in params]
        for r in range(3):
            rows = B.shard_rows(n, 3, r)
            lo, hi = B.pair_offset(n, rows[0]), B.pair_offset(n, rows[1])
            part = emb.fused_objective(fn, target[lo:hi], None, rows=rows, **kw)
            pg = torch.autograd.grad(part, params)
            tot += part.item()
            gsum = [s + p for s, p in zip(gsum, pg)]
        assert abs(tot - ref.item()) <= tol * abs(ref.item())
        for a, b in zip(gsum, rg):
            err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
            assert err <= 20 * tol, err


def test_fused_loss_full_size_tall_tiles():
    """At >= 8.4 M pairs the backward / fused-loss kernels switch to 16-row tiles: same numbers as the
    unfused path (forward kernel + framework loss + 16-row backward) at n = 4200, SPD(3) fp32."""
    from graphembed.manifolds import SymmetricPositiveDefinite as SPD
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import QuotientLoss, StressLoss
    n = 4200
    torch.manual_seed(8)
    with torch.device('cuda'):
        emb = ManifoldEmbedding(n, [SPD(3)])
        target = torch.rand(n * (n - 1) // 2) * 0.05 + 0.02
    params = [emb.xs[0], emb.scales[0]]
    for fn, kw in ((StressLoss(), {}), (QuotientLoss(), dict(epoch=2, alpha=1.1))):
        ref = fn(target, emb.compute_dists(None), **kw)
        rg = torch.autograd.grad(ref, params)
        loss = emb.fused_objective(fn, target, None, **kw)
        g = torch.autograd.grad(loss, params)
        assert abs(loss.item() - ref.item()) <= 5e-5 * abs(ref.item()), (loss.item(), ref.item())
        err = (g[0] - rg[0]).abs().max().item() / rg[0].abs().max().item()
        assert err <= 2e-4, err
        assert abs(g[1].item() - rg[1].item()) <= 2e-3 * abs(rg[1].item())


def test_graphed_minibatch_steps_match_eager():
    """Node-minibatch training (train.py:198-222, batch_size 512 in the paper grid) as a replayed HIP graph:
    the index buffer is refreshed in place between replays; targets come from mm_pair_gather, the
    embedding rows through the index_add backward.  Same parameters as the eager loop."""
    from graphembed import manifolds as M
    from graphembed.data import GraphDataset
    from graphembed.graphed import GraphedTrainStep
    from graphembed.modules import BatchedObjective, ManifoldEmbedding
    from graphembed.objectives import StressLoss
    from graphembed.optim import RiemannianSGD
    n, bs = 900, 256
    torch.set_default_dtype(torch.float64)
    try:
        def build():
            torch.manual_seed(9)
            with torch.device('cuda'):
                emb = ManifoldEmbedding(n, [M.SymmetricPositiveDefinite(3)])
                ds = GraphDataset(torch.rand(n * (n - 1) // 2) + 0.1)
            obj = BatchedObjective(StressLoss(), ds, emb)
            opts = [RiemannianSGD(list(emb.xs), lr=0.01, exact=True, max_grad_norm=20),
                    RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)]
            return emb, obj, opts
        torch.manual_seed(3)
        perm = torch.randperm(n, device='cuda')
        batches = [perm[k * bs:(k + 1) * bs] for k in range(3)] * 2
        emb_e, obj_e, opts_e = build()
        for idx in batches:
            for o in opts_e:
                o.zero_grad()
            obj_e(idx).backward()
            for o in opts_e:
                o.step()
        emb_g, obj_g, opts_g = build()
        idx_static = batches[0].clone()
        step = GraphedTrainStep(lambda: obj_g(idx_static), opts_g, warmup=1).capture()  # = batches[0]
        for idx in batches[1:]:
            idx_static.copy_(idx)
            step()
        np.testing.assert_allclose(emb_g.xs[0].detach().cpu().numpy(), emb_e.xs[0].detach().cpu().numpy(),
                                   rtol=1e-8, atol=1e-10)
        assert abs(emb_g.scales[0].item() - emb_e.scales[0].item()) <= 1e-10
        # the gather kernel against the dense fancy-index of the reference
        sub = obj_e.dataset.pdists[batches[0]][:, batches[0]]
        iu = torch.triu_indices(bs, bs, 1, device='cuda')
        assert torch.equal(obj_e.dataset[batches[0]], sub[iu[0], iu[1]])
    finally:
        torch.set_default_dtype(torch.float32)


@pytest.mark.parametrize('dname', ['f32', 'f64'])
@pytest.mark.parametrize('loss_name', ['stress', 'quotient'])
def test_product_minibatch_inside_pair_kernel(dname, loss_name):
    """Node minibatch of a product embedding handled entirely by the mixed-manifold pair kernel
    (mm_product_pairs_loss_subset: rows, targets and gradient rows addressed through the index vector)
    == gather rows -> compute_dists -> objective(dataset[idx]) -> autograd (modules.py:84-105,
    data/dataset.py:19-27), full-size gradients incl. the zero rows; shards of the batch's pair list sum up;
    and as a replayed graph with the index buffer refreshed in place it tracks the eager loop."""
    from graphembed import _backend as B
    from graphembed import manifolds as M
    from graphembed.data import GraphDataset
    from graphembed.graphed import GraphedTrainStep
    from graphembed.modules import BatchedObjective, ManifoldEmbedding
    from graphembed.objectives import QuotientLoss, StressLoss
    from graphembed.optim import RiemannianSGD
    dt = {'f32': torch.float32, 'f64': torch.float64}[dname]
    n, bs = 700, 200
    fn, kw = (StressLoss(), {}) if loss_name == 'stress' else (QuotientLoss(), dict(epoch=1, alpha=1.2))
    tol = 2e-4 if dname == 'f32' else 1e-10
    torch.set_default_dtype(dt)
    try:
        def build():
            torch.manual_seed(4)
            with torch.device('cuda'):
                emb = ManifoldEmbedding(n, [M.Lorentz(6), M.Euclidean(3), M.Sphere(6), M.SymmetricPositiveDefinite(2)])
                with torch.no_grad():
                    emb.perturb(0.3)
                ds = GraphDataset(torch.rand(n * (n - 1) // 2) * 3 + 0.5)
            return emb, ds
        emb, ds = build()
        params = list(emb.xs) + list(emb.scales)
        idx = torch.randperm(n, device='cuda')[:bs]
        # reference-shaped path
        ref = fn(ds[idx], emb.compute_dists(idx), **kw)
        rg = torch.autograd.grad(ref, params)
        lib, calls = B.lib(), []
        orig = lib.call
        lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            loss = BatchedObjective(fn, ds, emb)(idx, **kw)
        finally:
            del lib.call
        assert calls == ['mm_product_pairs_loss_subset'], calls
        g = torch.autograd.grad(loss, params)
        assert abs(loss.item() - ref.item()) <= tol * abs(ref.item())
        for a, b in zip(g, rg):
            assert a.shape == b.shape
            err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
            assert err <= 50 * tol, err
        rest = torch.ones(n, dtype=torch.bool, device='cuda')
        rest[idx] = False
        for a in g[:4]:
            assert not a[rest].any()          # rows outside the batch: exactly zero
        tot, gsum = 0.0, [torch.zeros_like(p) for p in params]
        for r in range(3):
            rows = B.shard_rows(bs, 3, r)
            part = emb.fused_objective(fn, None, idx, rows=rows, dense=ds.pdists, **kw)
            pg = torch.autograd.grad(part, params)
            tot += part.item()
            gsum = [s_ + p for s_, p in zip(gsum, pg)]
        assert abs(tot - ref.item()) <= tol * abs(ref.item())
        for a, b in zip(gsum, rg):
            err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
            assert err <= 50 * tol, err
        if loss_name != 'stress':
            return
        # graph replay over several batches == eager loop
        perm = torch.randperm(n, device='cuda')
        batches = [perm[k * bs:(k + 1) * bs] for k in range(3)] * 2

        def trainer():
            emb_, ds_ = build()
            obj = BatchedObjective(fn, ds_, emb_)
            opts = [RiemannianSGD(list(emb_.xs), lr=0.01, exact=True, max_grad_norm=20),
                    RiemannianSGD(list(emb_.scales), lr=1e-4, max_grad_norm=500)]
            return emb_, obj, opts
        emb_e, obj_e, opts_e = trainer()
        for b in batches:
            for o in opts_e:
                o.zero_grad()
            obj_e(b).backward()
            for o in opts_e:
                o.step()
        emb_g, obj_g, opts_g = trainer()
        idx_static = batches[0].clone()
        step = GraphedTrainStep(lambda: obj_g(idx_static), opts_g, warmup=1).capture()
        for b in batches[1:]:
            idx_static.copy_(b)
            step()
        for a, b in zip(emb_g.xs, emb_e.xs):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(),
                                       rtol=1e-3 if dname == 'f32' else 1e-8, atol=1e-5 if dname == 'f32' else 1e-10)
    finally:
        torch.set_default_dtype(torch.float32)


@pytest.mark.parametrize('case', ['spd3', 'lorentz', 'product', 'product_perfactor'])
def test_quotient_loss_schedule_in_device_memory(case):
    """QuotientLoss.on_device(): {alpha, eps = 1/(epoch+1)} are read from device memory by every fused loss
    kernel, so ONE captured graph follows the per-epoch schedule (objectives.py:16-36, train.py:170-178):
    replayed with set_epoch() it tracks the eager loop that passes epoch/alpha by value."""
    from graphembed import manifolds as M
    from graphembed.graphed import GraphedTrainStep
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import QuotientLoss
    from graphembed.optim import RiemannianSGD
    n = 150
    mans = {'spd3': lambda: [M.SymmetricPositiveDefinite(3)], 'lorentz': lambda: [M.Lorentz(5)],
            'product': lambda: [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)],
            'product_perfactor': lambda: [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)]}[case]
    torch.set_default_dtype(torch.float64)
    try:
        def build():
            torch.manual_seed(5)
            with torch.device('cuda'):
                emb = ManifoldEmbedding(n, mans())
                with torch.no_grad():
                    emb.perturb(0.2)
            emb.pair_kernel = case != 'product_perfactor'
            opts = [RiemannianSGD(list(emb.xs), lr=1e-4, exact=True, max_grad_norm=20),
                    RiemannianSGD(list(emb.scales), lr=1e-5, max_grad_norm=500)]
            return emb, opts
        torch.manual_seed(1)
        target = torch.rand(n * (n - 1) // 2, device='cuda') * 0.9 + 0.1
        alphas = [1.0, 1.0, 1.3, 1.3, 0.8, 0.8]
        emb_e, opts_e = build()
        fn_e = QuotientLoss()
        losses_e = []
        for epoch, alpha in enumerate(alphas):
            for o in opts_e:
                o.zero_grad()
            loss = emb_e.fused_objective(fn_e, target, None, epoch=epoch, alpha=alpha)
            loss.backward()
            for o in opts_e:
                o.step()
            losses_e.append(loss.item())
        emb_g, opts_g = build()
        fn_g = QuotientLoss()
        fn_g.on_device('cuda')
        fn_g.set_epoch(0, alphas[0])
        # the by-value epoch/alpha of the recorded call are deliberately wrong: the device copy counts
        step = GraphedTrainStep(lambda: emb_g.fused_objective(fn_g, target, None, epoch=0, alpha=alphas[0]),
                                opts_g, warmup=1).capture()
        losses_g = [step.warmup_losses[0].item()]
        for epoch in range(1, len(alphas)):
            fn_g.set_epoch(epoch, alphas[epoch])
            losses_g.append(step().item())
        np.testing.assert_allclose(losses_g, losses_e, rtol=1e-9)
        assert len(set(np.round(losses_e, 6))) == len(losses_e)   # the schedule does change the loss
        for a, b in zip(emb_g.xs, emb_e.xs):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-8, atol=1e-10)
    finally:
        torch.set_default_dtype(torch.float32)


@pytest.mark.parametrize('mix', ['spd3', 'product'])
def test_sharded_fused_objective_sums_to_full(mix):
    """graphembed.parallel.sharded_fused_objective: each rank's fused loss+gradient pass over its pair rows
    (here the 3 ranks of a world are played one after the other; the all-reduce is the sum) gives the full
    loss and the full gradients of compute_dists -> objective -> backward."""
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import QuotientLoss
    from graphembed.parallel import PairShard, sharded_fused_objective
    n = 260
    torch.manual_seed(8)
    torch.set_default_dtype(torch.float64)
    try:
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, [M.SymmetricPositiveDefinite(3)] if mix == 'spd3'
                                    else [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)])
            with torch.no_grad():
                emb.perturb(0.3)
            target = torch.rand(n * (n - 1) // 2) * 2 + 0.2
    finally:
        torch.set_default_dtype(torch.float32)
    fn, kw = QuotientLoss(), dict(epoch=2, alpha=1.1)
    params = list(emb.xs) + list(emb.scales)
    ref = fn(target, emb.compute_dists(None), **kw)
    rg = torch.autograd.grad(ref, params)
    tot, gsum = 0.0, [torch.zeros_like(p) for p in params]
    for r in range(3):
        shard = PairShard(n, world=3, rank=r)
        part = sharded_fused_objective(emb, fn, target, shard, **kw)
        pg = torch.autograd.grad(part, params)
        tot += part.item()
        gsum = [s_ + p for s_, p in zip(gsum, pg)]
    assert abs(tot - ref.item()) <= 1e-9 * abs(ref.item())
    for a, b in zip(gsum, rg):
        assert (a - b).abs().max().item() <= 1e-8 * max(b.abs().max().item(), 1e-30)


def test_graphed_step_follows_a_learning_rate_schedule():
    """A scheduler lowers the learning rate (ReduceLROnPlateau, train.py:174): GraphedTrainStep notices the
    changed by-value hyper-parameter, runs that step eagerly and re-records — same trajectory as the eager
    loop with the same schedule."""
    from graphembed import manifolds as M
    from graphembed.graphed import GraphedTrainStep
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import StressLoss
    from graphembed.optim import RiemannianSGD
    n = 120
    torch.set_default_dtype(torch.float64)
    try:
        def build():
            torch.manual_seed(6)
            with torch.device('cuda'):
                emb = ManifoldEmbedding(n, [M.Lorentz(4), M.SymmetricPositiveDefinite(2)])
            return emb, [RiemannianSGD(list(emb.xs), lr=0.02, exact=True, max_grad_norm=20),
                         RiemannianSGD(list(emb.scales), lr=1e-3, max_grad_norm=500)]
        torch.manual_seed(2)
        target = torch.rand(n * (n - 1) // 2, device='cuda') * 0.9 + 0.1
        fn = StressLoss()
        lrs = [0.02, 0.02, 0.02, 0.002, 0.002, 0.0002, 0.0002]
        emb_e, opts_e = build()
        losses_e = []
        for lr in lrs:
            opts_e[0].param_groups[0]['lr'] = lr
            for o in opts_e:
                o.zero_grad()
            loss = emb_e.fused_objective(fn, target, None)
            loss.backward()
            for o in opts_e:
                o.step()
            losses_e.append(loss.item())
        emb_g, opts_g = build()
        step = GraphedTrainStep(lambda: emb_g.fused_objective(fn, target, None), opts_g, warmup=1).capture()
        losses_g = [step.warmup_losses[0].item()]
        graphs = [step.graph]          # (kept alive: ids of freed graph objects get reused)
        for lr in lrs[1:]:
            opts_g[0].param_groups[0]['lr'] = lr
            losses_g.append(step().item())
            if step.graph is not graphs[-1]:
                graphs.append(step.graph)
        np.testing.assert_allclose(losses_g, losses_e, rtol=1e-9)
        assert len(graphs) == 3                     # recorded once per learning rate
        for a, b in zip(emb_g.xs, emb_e.xs):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-8, atol=1e-10)
    finally:
        torch.set_default_dtype(torch.float32)


def test_graphed_step_unrolled():
    """GraphedTrainStep(unroll=3): three consecutive steps per recorded graph — same trajectory as the eager loop,
    all three losses available after a replay."""
    from graphembed import manifolds as M
    from graphembed.graphed import GraphedTrainStep
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import StressLoss
    from graphembed.optim import RiemannianAdam, RiemannianSGD
    n = 90
    torch.set_default_dtype(torch.float64)
    try:
        def build():
            torch.manual_seed(7)
            with torch.device('cuda'):
                emb = ManifoldEmbedding(n, [M.Lorentz(5), M.Sphere(4), M.SymmetricPositiveDefinite(2)])
            return emb, [RiemannianSGD(list(emb.xs), lr=0.01, exact=True, max_grad_norm=20),
                         RiemannianAdam(list(emb.scales), lr=1e-3)]
        torch.manual_seed(3)
        target = torch.rand(n * (n - 1) // 2, device='cuda') * 0.9 + 0.1
        fn = StressLoss()
        emb_e, opts_e = build()
        losses_e = []
        for _ in range(7):
            for o in opts_e:
                o.zero_grad()
            loss = emb_e.fused_objective(fn, target, None)
            loss.backward()
            for o in opts_e:
                o.step()
            losses_e.append(loss.item())
        emb_g, opts_g = build()
        step = GraphedTrainStep(lambda: emb_g.fused_objective(fn, target, None), opts_g, warmup=1, unroll=3).capture()
        losses_g = [step.warmup_losses[0].item()]
        for _ in range(2):
            last = step()
            losses_g += [l.item() for l in step.losses]
            assert last is step.losses[-1]
        np.testing.assert_allclose(losses_g, losses_e, rtol=1e-9)
        for a, b in zip(emb_g.xs, emb_e.xs):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-8, atol=1e-10)
    finally:
        torch.set_default_dtype(torch.float32)


@pytest.mark.parametrize('opt_name', ['rsgd', 'radam'])
def test_training_engine_flow(opt_name):
    """The sequence of calls the reference's TrainingEngine makes (train.py:104-222, experiments/run_grid.py:24-36) —
    burn-in with frozen scales, CPU randperm minibatches with `drop_last_n`, zero_grad / backward / step on two
    optimizers, ReduceLROnPlateau, deepcopy of the best embedding, perturb, stabilize, validation on compute_dists,
    state_dict round trip — runs on this package's classes unchanged and trains."""
    import copy
    from graphembed import manifolds as M
    from graphembed.data import GraphDataset
    from graphembed.metrics import pearsonr
    from graphembed.modules import BatchedObjective, ManifoldEmbedding
    from graphembed.objectives import QuotientLoss
    from graphembed.optim import RiemannianAdam, RiemannianSGD
    torch.manual_seed(0)
    n, bs, drop_last_n = 70, 32, 5
    torch.set_default_dtype(torch.float64)
    try:
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, [M.Lorentz(4), M.Sphere(3), M.SymmetricPositiveDefinite(2)])
            pts = torch.randn(n, 3)
            ds = GraphDataset(torch.pdist(pts))          # targets: squared, max-normalised (dataset.py:9-13)
        Opt = RiemannianSGD if opt_name == 'rsgd' else RiemannianAdam
        opts = [Opt(list(emb.xs), lr=0.05, exact=True, max_grad_norm=20),
                Opt(list(emb.curvature_params), lr=1e-3, max_grad_norm=500)]
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opts[0], factor=0.1, patience=2, threshold=1e-4, min_lr=1e-5)
        obj = BatchedObjective(QuotientLoss(), ds, emb)

        def train(alpha, epoch):
            perm = torch.randperm(n)                       # CPU indices, as the reference draws them
            total = 0.0
            for i in range(0, n, bs):
                idx = perm[i:i + bs]
                if len(idx) < drop_last_n:
                    break
                loss = obj(idx, alpha=alpha, epoch=epoch).sum()
                for o in opts:
                    o.zero_grad()
                loss.backward()
                for o in opts:
                    o.step()
                total += loss.item()
            return total
        with torch.no_grad():
            r0 = pearsonr(emb.compute_dists(None), ds[None]).item()
        emb.burnin(True)                                   # scales frozen during burn-in (train.py:140-168)
        s0 = [s.item() for s in emb.scales]
        for epoch in range(1, 3):
            train(0.5, epoch)
        assert [s.item() for s in emb.scales] == s0
        emb.burnin(False)
        best = dict(loss=1e30, embedding=copy.deepcopy(emb))
        losses = []
        for epoch in range(1, 21):
            loss = train(1.0, epoch)
            sched.step(loss)
            losses.append(loss)
            if loss < best['loss']:
                best = dict(loss=loss, embedding=copy.deepcopy(emb))
            if epoch % 5 == 0:
                with torch.no_grad():
                    emb.perturb(0.05 / epoch)
            if epoch % 4 == 0:
                with torch.no_grad():
                    emb.stabilize()
        assert all(np.isfinite(losses))   # (the quotient loss is not comparable across epochs: eps = 1/(epoch+1))
        with torch.no_grad():
            r = pearsonr(emb.compute_dists(None), ds[None]).item()
        assert r > max(0.5, r0 + 0.2), (r0, r)
        # the snapshot is an independent, fully functional embedding (manifold tags kept)
        snap = best['embedding']
        assert all(a.manifold is not None and a.data_ptr() != b.data_ptr() for a, b in zip(snap.xs, emb.xs))
        assert torch.isfinite(snap.compute_dists(None)).all()
        # state_dict round trip into a fresh embedding (keys xs.k / scales.k)
        with torch.device('cuda'):
            fresh = ManifoldEmbedding(n, [M.Lorentz(4), M.Sphere(3), M.SymmetricPositiveDefinite(2)])
        fresh.load_state_dict(emb.state_dict())
        assert sorted(emb.state_dict()) == ['scales.0', 'scales.1', 'scales.2', 'xs.0', 'xs.1', 'xs.2']
        assert torch.equal(fresh.compute_dists(None), emb.compute_dists(None))
        osd = opts[0].state_dict()
        opts[0].load_state_dict(osd)
        train(1.0, 21)
    finally:
        torch.set_default_dtype(torch.float32)


# ---- config 4 at its real size through the kernel BASELINE.json names for it (round-2 review, weak 1a) ------------
def _exact_product_objective(xs, kinds, scales, target_full, n, pairs, loss):
    """loss, d loss / d x_k, d loss / d scale_k of ManifoldEmbedding.compute_dists (modules.py:84-88) + objectives.py
    (stress: 39-45; quotient: 16-36 with both terms) in fp64 from oracle/exact.c's per-factor distances and gradients:
    m = sum_k softplus(s_k) d2_k restricted to `pairs` (indices into the full pair vector of the n points)."""
    from oracle import exact
    sp = [np.log1p(np.exp(s)) for s in scales]
    d2 = []
    for x, kind in zip(xs, kinds):
        d2.append(exact.spd_pdist(x) if kind == 'spd' else exact.vec_pdist(kind, x))
    m = sum(s * d for s, d in zip(sp, d2))
    t = target_full
    if loss == 'stress':
        lt, dl = (m - t) ** 2, 2 * (m - t)
    else:
        alpha, eps = 1.0, 0.25
        q1, q2 = m / (alpha * t) - 1, alpha * t / (m + eps) - 1
        lt = np.abs(q1) + np.abs(q2)
        dl = np.sign(q1) / (alpha * t) - np.sign(q2) * alpha * t / (m + eps) ** 2
    mask = np.zeros_like(m)
    mask[pairs] = 1.0
    lval = float((lt * mask).sum())
    grads, gscale = [], []
    for x, kind, s, s_raw, d in zip(xs, kinds, sp, scales, d2):
        up = dl * mask * s
        grads.append(exact.spd_pdist_grad(x, up) if kind == 'spd' else exact.vec_pdist_grad(kind, x, up))
        gscale.append(float((dl * mask * d).sum() / (1 + np.exp(-s_raw))))
    return lval, grads, gscale


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64], ids=['f32', 'f64'])
@pytest.mark.parametrize('loss', ['stress', 'quotient'])
def test_config4_product_pair_kernel_full_size(dtype, loss):
    """csphd (n = 1025) -> H^5 x S^5 x SPD(2): mm_product_pairs_loss — loss, the gradients of all three factors and the
    three scale gradients — against oracle/exact.c per factor combined on the host, and mm_product_pairs_loss_subset on a
    512-node slice of a randperm (train.py:206-209) against the same checker on the gathered points."""
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import QuotientLoss, StressLoss
    from oracle import ref_port as rp
    n = 1025
    gen = torch.Generator().manual_seed(11)
    kinds = ['lorentz', 'sphere', 'spd']
    x64 = [rp.make('lorentz', 6).rand(n, ir=0.3, dtype=torch.float64, generator=gen),
           rp.make('sphere', 6).rand(n, ir=0.3, dtype=torch.float64, generator=gen),
           spd_points(n, 2, 5, 'wide')]
    raw = [0.5, 0.3, 0.7]
    P = n * (n - 1) // 2
    target = torch.rand(P, dtype=torch.float64, generator=gen) * 0.9 + 0.05
    xin = [x.to(dtype) for x in x64]
    tin = target.to(dtype)
    fn = StressLoss() if loss == 'stress' else QuotientLoss()
    kw = {} if loss == 'stress' else {'epoch': 3, 'alpha': 1.0}       # eps = 1 / (epoch + 1) = 0.25
    f32 = dtype == torch.float32
    rtol_l, gtol = (2e-5, 3e-4) if f32 else (1e-11, 1e-9)

    def embedding(points):
        with torch.device('cuda'):
            emb = ManifoldEmbedding(points[0].shape[0], [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)])
        with torch.no_grad():
            for p, x in zip(emb.xs, points):
                p.copy_(x.cuda())
            for s, v in zip(emb.scales, raw):
                s.fill_(v)
        return emb

    def compare(emb, lossv, ref, what):
        lref, gref, sref = ref
        assert abs(lossv - lref) <= rtol_l * abs(lref), (what, lossv, lref)
        for k, (p, g) in enumerate(zip(emb.xs, gref)):
            err = np.abs(p.grad.double().cpu().numpy() - g).max() / np.abs(g).max()
            assert err <= gtol, f'{what}: factor {kinds[k]} grad rel err {err:.3e}'
        for k, (s, g) in enumerate(zip(emb.scales, sref)):
            assert abs(s.grad.item() - g) <= (2e-4 if f32 else 1e-9) * max(abs(g), 1e-3 * abs(lref)), (what, k, s.grad.item(), g)

    torch.set_default_dtype(dtype)
    try:
        # full batch: all pairs of the 1025 nodes through the mixed-manifold pair kernel
        emb = embedding(xin)
        assert emb.pair_kernel
        lossv = emb.fused_objective(fn, tin.cuda(), None, **kw)
        lossv.backward()
        ref = _exact_product_objective([x.double().numpy() for x in xin], kinds, raw, tin.double().numpy(), n,
                                       np.arange(P), loss)
        compare(emb, lossv.item(), ref, f'config 4 full batch {loss} {dtype}')
        # node minibatch of 512 inside the pair kernel (index vector addresses rows, dense targets and gradient rows)
        from graphembed.data import GraphDataset
        from graphembed.modules import BatchedObjective
        emb = embedding(xin)
        ds = GraphDataset(tin.cuda().sqrt())
        ds.condensed, ds._dense = tin.cuda(), None      # (the constructor normalises by the maximum: set the targets as they are)
        obj = BatchedObjective(fn, ds, emb)
        idx = torch.randperm(n, generator=gen)[:512]
        lossv = obj(idx.cuda(), **kw)
        lossv.backward()
        sub = [x[idx].double().numpy() for x in xin]
        i, j = torch.triu_indices(512, 512, 1)
        a, b = torch.minimum(idx[i], idx[j]), torch.maximum(idx[i], idx[j])
        tsub = tin.double()[a * (2 * n - a - 1) // 2 + (b - a - 1)].numpy()
        lref, gsub, sref = _exact_product_objective(sub, kinds, raw, tsub, 512, np.arange(512 * 511 // 2), loss)
        gfull = []
        for g, x in zip(gsub, xin):
            full = np.zeros(x.shape)
            full[idx.numpy()] = g
            gfull.append(full)
        compare(emb, lossv.item(), (lref, gfull, sref), f'config 4 minibatch 512 {loss} {dtype}')
    finally:
        torch.set_default_dtype(torch.float32)
