"""The node-minibatch objective (graphembed/train.py:206-213, modules.py:84-105, data/dataset.py:19-27) pinned DIRECTLY to values
recorded from the real reference (tests/golden/gen_golden_minibatch.py): loss and all gradients of
`objective_fn(dataset[idx], embedding.compute_dists(idx))` — against the oracle port on the CPU, and on the GPU against the
in-kernel minibatch path (`mm_product_pairs_loss_subset`: rows, targets and gradient rows addressed through the index
vector) as well as the gather -> compute_dists -> objective path."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'minibatch.npz')))
# round 4: single factors whose minibatches run inside their own pair kernels (gen_golden_minibatch.py --wide)
GOLD.update(np.load(os.path.join(ROOT, 'tests', 'golden', 'minibatch_wide.npz')))
DT = {'f32': torch.float32, 'f64': torch.float64}
WIDE = ['spd4', 'spd6', 'lorentz24', 'sphere20', 'euclidean40']
CASES = ['product', 'spd3', 'lorentz11'] + WIDE
LOSSES = {'stress': {}, 'quotient': dict(epoch=2, alpha=1.3)}


def manifolds_of(case, M, spd):
    return {'product': lambda: [M.Lorentz(6), M.Sphere(6), spd(2)], 'spd3': lambda: [spd(3)], 'lorentz11': lambda: [M.Lorentz(11)],
            'spd4': lambda: [spd(4)], 'spd6': lambda: [spd(6)], 'lorentz24': lambda: [M.Lorentz(24)],
            'sphere20': lambda: [M.Sphere(20)], 'euclidean40': lambda: [M.Euclidean(40)]}[case]()


SPD_FACTOR = {'product': 2, 'spd3': 0, 'spd4': 0, 'spd6': 0}


def sym_if_spd(a, is_spd):
    return 0.5 * (a + np.swapaxes(a, -1, -2)) if is_spd else a


def check(loss, grads, base, lname, nfac, spd_factor, tol):
    ref = float(GOLD[f'{base}/{lname}/loss'])
    assert abs(float(loss) - ref) <= tol * abs(ref), (float(loss), ref)
    for k in range(nfac):
        a = sym_if_spd(grads[k], k == spd_factor)
        b = sym_if_spd(GOLD[f'{base}/{lname}/grad_x_{k}'].astype(np.float64), k == spd_factor)
        assert a.shape == b.shape
        assert np.abs(a - b).max() <= 20 * tol * max(np.abs(b).max(), 1e-30), (lname, k, np.abs(a - b).max(), np.abs(b).max())
        gs, rs = float(grads[nfac + k]), float(GOLD[f'{base}/{lname}/grad_s_{k}'])
        assert abs(gs - rs) <= 20 * tol * max(abs(rs), 1e-30), (lname, 'scale', k, gs, rs)


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dname', list(DT))
def test_oracle_port_minibatch_objective(case, dname):
    from oracle import ref_port as rp
    mans = manifolds_of(case, rp, rp.SPD)
    base = f'{case}/{dname}'
    dt = DT[dname]
    idx = torch.from_numpy(GOLD[f'{base}/idx'])
    sq = torch.from_numpy(GOLD[f'{base}/graph_d']).to(dt).pow(2)
    sq = sq / sq.max()                                  # GraphDataset.__init__, data/dataset.py:10-13
    n = 61
    dense = torch.zeros(n, n, dtype=dt)
    i, j = torch.triu_indices(n, n, 1)
    dense[i, j] = sq
    dense = dense + dense.T
    sub = dense[idx][:, idx]
    bi, bj = torch.triu_indices(len(idx), len(idx), 1)
    gd = sub[bi, bj]                                     # dataset[idx]
    for lname, kw in LOSSES.items():
        xs = [torch.from_numpy(GOLD[f'{base}/x_{k}']).requires_grad_() for k in range(len(mans))]
        sc = [torch.tensor(float(s), dtype=dt, requires_grad=True) for s in GOLD[f'{base}/scales']]
        md = rp.compute_dists(mans, xs, sc, idx)
        loss = rp.stress_loss(gd, md) if lname == 'stress' else rp.quotient_loss(gd, md, **kw)
        grads = [g.double().numpy() for g in torch.autograd.grad(loss, xs + sc)]
        spd_factor = SPD_FACTOR.get(case, -1)
        check(loss.item(), grads, base, lname, len(mans), spd_factor, 2e-4 if dname == 'f32' else 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dname', list(DT))
@pytest.mark.parametrize('path', ['in_kernel', 'gather'])
def test_minibatch_objective_gpu(case, dname, path):
    from graphembed import _backend as B
    from graphembed import manifolds as M
    from graphembed.data import GraphDataset
    from graphembed.modules import BatchedObjective, ManifoldEmbedding
    from graphembed.objectives import QuotientLoss, StressLoss
    dt = DT[dname]
    base = f'{case}/{dname}'
    mans = manifolds_of(case, M, M.SymmetricPositiveDefinite)
    spd_factor = SPD_FACTOR.get(case, -1)
    torch.set_default_dtype(dt)
    try:
        with torch.device('cuda'):
            emb = ManifoldEmbedding(61, mans)
            ds = GraphDataset(torch.from_numpy(GOLD[f'{base}/graph_d']).to(dt).cuda())
        with torch.no_grad():
            for k, x in enumerate(emb.xs):
                x.copy_(torch.from_numpy(GOLD[f'{base}/x_{k}']).cuda())
            for k, s in enumerate(emb.scales):
                s.fill_(float(GOLD[f'{base}/scales'][k]))
        idx = torch.from_numpy(GOLD[f'{base}/idx']).cuda()
        params = list(emb.xs) + list(emb.scales)
        for lname, kw in LOSSES.items():
            fn = StressLoss() if lname == 'stress' else QuotientLoss()
            if path == 'in_kernel':
                lib, calls = B.lib(), []
                orig = lib.call
                lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
                try:
                    loss = BatchedObjective(fn, ds, emb)(idx, **kw)
                finally:
                    del lib.call
                # the index vector goes into the kernel: the mixed-manifold one, or the factor's own (SPD(4...9), m > 16)
                want = {'spd4': 'mm_spd_pdist_loss_subset', 'spd6': 'mm_spd_pdist_loss_subset'}.get(
                    case, 'mm_vec_pdist_loss_subset' if case in WIDE else 'mm_product_pairs_loss_subset')
                assert want in calls and not any('gather' in c for c in calls), calls
            else:
                loss = fn(ds[idx], emb.compute_dists(idx), **kw)
            grads = [g.double().cpu().numpy() for g in torch.autograd.grad(loss, params)]
            # fp64: the reference's SPD closed forms carry their eps terms (1e-8 ... 1e-6 of bias); Lorentz has none
            tol64 = 5e-6 if spd_factor >= 0 else 1e-9
            check(loss.item(), grads, base, lname, len(mans), spd_factor, 2e-4 if dname == 'f32' else tol64)
            rest = np.ones(61, dtype=bool)
            rest[GOLD[f'{base}/idx']] = False
            for k in range(len(mans)):
                assert not grads[k][rest].any()          # rows outside the batch: exactly zero, as autograd gives them
    finally:
        torch.set_default_dtype(torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['spd3', 'spd4', 'spd6', 'lorentz24', 'euclidean40'])
@pytest.mark.parametrize('dname', list(DT))
def test_native_minibatch_step_matches_the_eager_loop(case, dname):
    """mm_train_step with batch_idx (the one-call step over a node minibatch) against the eager loop on the same batches:
    BatchedObjective(idx) -> backward -> optimizer.step(), three consecutive batches, RSGD and Adam; every point is stepped
    (zero gradient outside the batch), as the reference's dense x.grad has it (train.py:206-222)."""
    from graphembed import manifolds as M
    from graphembed.data import GraphDataset
    from graphembed.modules import BatchedObjective, ManifoldEmbedding
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import QuotientLoss, StressLoss
    from graphembed.optim import RiemannianAdam, RiemannianSGD
    dt = DT[dname]
    base = f'{case}/{dname}'
    torch.set_default_dtype(dt)
    try:
        for opt_name, fn, kw in (('rsgd', StressLoss(), {}), ('adam', QuotientLoss(), dict(epoch=2, alpha=1.3))):
            runs = []
            for native in (False, True):
                with torch.device('cuda'):
                    emb = ManifoldEmbedding(61, manifolds_of(case, M, M.SymmetricPositiveDefinite))
                    ds = GraphDataset(torch.from_numpy(GOLD[f'{base}/graph_d']).to(dt).cuda())
                with torch.no_grad():
                    emb.xs[0].copy_(torch.from_numpy(GOLD[f'{base}/x_0']).cuda())
                    emb.scales[0].fill_(float(GOLD[f'{base}/scales'][0]))
                if opt_name == 'rsgd':
                    opts = [RiemannianSGD(list(emb.xs), lr=0.05, exact=True, max_grad_norm=20), RiemannianSGD(list(emb.scales), lr=1e-3, max_grad_norm=500)]
                else:
                    opts = [RiemannianAdam(list(emb.xs), lr=0.01, exact=True, max_grad_norm=20), RiemannianAdam(list(emb.scales), lr=1e-3)]
                gen = torch.Generator().manual_seed(5)
                batches = [torch.randperm(61, generator=gen)[:23].cuda() for _ in range(3)]
                losses = []
                if native:
                    step = NativeTrainStep(emb, fn, None, opts, dense=ds.pdists)
                    for idx in batches:
                        losses.append(float(step(indices=idx, **kw)))
                else:
                    obj = BatchedObjective(fn, ds, emb)
                    for idx in batches:
                        loss = obj(idx, **kw)
                        for o in opts:
                            o.zero_grad(set_to_none=True)
                        loss.backward()
                        for o in opts:
                            o.step()
                        losses.append(float(loss))
                runs.append((losses, emb.xs[0].detach().double().cpu().numpy(), float(emb.scales[0])))
            (l0, x0, s0), (l1, x1, s1) = runs
            tol = 2e-4 if dname == 'f32' else 1e-9
            assert np.allclose(l0, l1, rtol=tol), (opt_name, l0, l1)
            assert np.abs(x0 - x1).max() <= tol * max(np.abs(x0).max(), 1.0), (opt_name, np.abs(x0 - x1).max())
            assert abs(s0 - s1) <= tol * max(abs(s0), 1.0)
    finally:
        torch.set_default_dtype(torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize('man_name,m', [('Euclidean', 5), ('Sphere', 3), ('Lorentz', 11), ('SPD', 3)])
def test_full_batch_step_behind_a_minibatch_step(man_name, m):
    """One NativeTrainStep alternating node minibatches and full batches (found by tools/fuzz_step.py, round 4): a vector
    factor's minibatch step takes the unfused kernels and does not rewrite the zero-padded copy of the points the fused
    full-batch step reads, so the stepper must prepare it again — it passed MM_WS_PREPARED and the full batch behind a
    minibatch ran on stale points (loss 641.2 against 630.7).  Against the eager loop, fp64."""
    import copy
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import StressLoss
    from graphembed.optim import RiemannianSGD
    n = 64
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(3)
        man = M.SymmetricPositiveDefinite(m) if man_name == 'SPD' else getattr(M, man_name)(m)
        with torch.device('cuda'):
            emb_a = ManifoldEmbedding(n, [man])
            with torch.no_grad():
                emb_a.perturb(0.3)
            target = torch.rand(n * (n - 1) // 2) * 0.9 + 0.05
        dense = torch.zeros(n, n, device='cuda')
        iu = torch.triu_indices(n, n, 1, device='cuda')
        dense[iu[0], iu[1]] = target
        dense = (dense + dense.t()).contiguous()
        emb_b = copy.deepcopy(emb_a)
        fn = StressLoss()
        opts = lambda e: [RiemannianSGD(list(e.xs), lr=1e-3, exact=True, max_grad_norm=20),  # noqa: E731
                          RiemannianSGD(list(e.scales), lr=1e-4, max_grad_norm=500)]
        oa, ob = opts(emb_a), opts(emb_b)
        step = NativeTrainStep(emb_b, fn, target, ob, dense=dense)
        gen = torch.Generator().manual_seed(11)
        plan = [None, torch.randperm(n, generator=gen)[:23].cuda(), None, torch.randperm(n, generator=gen)[:40].cuda(), None, None]
        for idx in plan:
            if idx is None:
                loss = emb_a.fused_objective(fn, target, None)
            else:
                loss = emb_a.fused_objective(fn, None, idx, dense=dense)
            for o in oa:
                o.zero_grad(set_to_none=True)
            loss.backward()
            for o in oa:
                o.step()
            got = step() if idx is None else step(indices=idx)
            assert abs(got.item() - loss.item()) <= 1e-9 * abs(loss.item()), (idx is None, got.item(), loss.item())
        assert (emb_a.xs[0] - emb_b.xs[0]).abs().max().item() <= 1e-10
    finally:
        torch.set_default_dtype(torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize('layout', ['spd4', 'spd3', 'lorentz24', 'euclidean5', 'product'])
@pytest.mark.parametrize('loss_name', ['stress', 'quotient'])
def test_degenerate_batches(layout, loss_name):
    """Batches of 0, 1 and 2 nodes through every in-kernel minibatch route (the mixed-manifold pair kernel, the single factors'
    own pair kernels) and through the one-call step: no pairs -> a zero loss with (dense) zero gradients, as the reference's sum
    over an empty pair list; two nodes -> the one pair's loss.  Round 4: an EMPTY index tensor has no storage, and the one-call
    step took its null pointer for "full batch" — it read the dense matrix as a pair vector and stepped the points on it."""
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import QuotientLoss, StressLoss
    from graphembed.optim import RiemannianSGD
    mans = {'spd4': lambda: [M.SymmetricPositiveDefinite(4)], 'spd3': lambda: [M.SymmetricPositiveDefinite(3)],
            'lorentz24': lambda: [M.Lorentz(24)], 'euclidean5': lambda: [M.Euclidean(5)],
            'product': lambda: [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)]}[layout]()
    fn = StressLoss() if loss_name == 'stress' else QuotientLoss()
    kw = dict(epoch=1, alpha=1.0)
    n = 37
    torch.manual_seed(5)
    with torch.device('cuda'):
        emb = ManifoldEmbedding(n, mans)
    dense = torch.rand(n, n, device='cuda')
    dense = (dense + dense.t()).contiguous()
    dense.fill_diagonal_(0)
    params = list(emb.xs) + list(emb.scales)
    for bs in (0, 1, 2):
        idx = torch.randperm(n, device='cuda')[:bs]
        loss = emb.fused_objective(fn, None, idx, dense=dense, **kw)
        assert loss is not None
        gs = torch.autograd.grad(loss, params)
        assert all(g.shape == p.shape and bool(torch.isfinite(g).all()) for g, p in zip(gs, params))
        if bs < 2:
            assert loss.item() == 0.0 and all(not g.any() for g in gs)
        else:
            iu = torch.triu_indices(bs, bs, 1, device='cuda')
            ref = fn(dense[idx][:, idx][iu[0], iu[1]], emb.compute_dists(idx), **kw)
            assert abs(loss.item() - ref.item()) <= 2e-5 * abs(ref.item())
    if len(mans) == 1:
        opts = [RiemannianSGD(list(emb.xs), lr=1e-3), RiemannianSGD(list(emb.scales), lr=1e-4)]
        step = NativeTrainStep(emb, fn, None, opts, dense=dense)
        before = emb.xs[0].detach().clone()
        for bs in (0, 1):
            got = step(indices=torch.randperm(n, device='cuda')[:bs], **kw)
            assert got.item() == 0.0
            assert torch.equal(emb.xs[0].detach(), before) or (emb.xs[0].detach() - before).abs().max().item() <= 1e-6   # (RSGD on zero gradients: exp(x, 0))
        got = step(indices=torch.randperm(n, device='cuda')[:2], **kw)
        assert got.item() > 0 and bool(torch.isfinite(emb.xs[0]).all())
        from graphembed import _backend as B
        with pytest.raises(B.BackendError):      # a batch size without an index vector is refused, not run as a full batch
            step._desc.batch_idx, step._desc.batch = None, 5
            B.lib().call('mm_train_step_run', __import__('ctypes').byref(step._desc), B.stream_of(dense))


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['spd4', 'lorentz24', 'spd3'])
def test_bad_minibatch_indices_are_caught_on_every_route(case):
    """The in-kernel minibatch routes address tables, dense targets and accumulators through the index vector (advisor, round 4:
    NativeTrainStep and the single-factor subset loss bypassed BatchedObjective's check).  HOST-side indices: out of range
    raises IndexError like the reference's x[idx] (modules.py:86) on every route; python-style negatives are wrapped; repeats go
    the gather / scatter way (BatchedObjective, fused_objective) or raise (NativeTrainStep has no such route).  DEVICE-side
    indices cannot be checked without a synchronisation: the kernels clamp node ids into the tables, so a bad index yields
    finite wrong numbers and no stray access."""
    from graphembed import manifolds as M
    from graphembed.data import GraphDataset
    from graphembed.modules import BatchedObjective, ManifoldEmbedding
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import StressLoss
    from graphembed.optim import RiemannianSGD
    n = 61
    base = f'{case}/f32'
    with torch.device('cuda'):
        emb = ManifoldEmbedding(n, manifolds_of(case, M, M.SymmetricPositiveDefinite))
        ds = GraphDataset(torch.from_numpy(GOLD[f'{base}/graph_d']).float().cuda())
    with torch.no_grad():
        emb.xs[0].copy_(torch.from_numpy(GOLD[f'{base}/x_0']).cuda())
    fn = StressLoss()
    obj = BatchedObjective(fn, ds, emb)
    opts = [RiemannianSGD(list(emb.xs), lr=0.01, exact=True, max_grad_norm=20), RiemannianSGD(list(emb.scales), lr=1e-3)]
    step = NativeTrainStep(emb, fn, None, opts, dense=ds.pdists)
    good = torch.tensor([3, 17, 40, 8, 59])
    for bad in (torch.tensor([3, 17, n, 8]), torch.tensor([3, -n - 1, 5])):
        with pytest.raises(IndexError):
            obj(bad)
        with pytest.raises(IndexError):
            emb.fused_objective(fn, None, bad, dense=ds.pdists)
        with pytest.raises(IndexError):
            step(indices=bad)
    # python-style negatives: the same batch as their wrapped form, on every route
    neg = torch.tensor([3, 17 - n, 40, 8 - n, -2])
    ref = float(obj(good))
    assert abs(float(obj(neg)) - ref) <= 1e-5 * abs(ref)
    assert abs(float(emb.fused_objective(fn, ds[good], neg)) - ref) <= 1e-5 * abs(ref)     # (gather route: targets given)
    # repeats: the reference accumulates the repeated row's gradient; the gather route does, the native step refuses
    rep = torch.tensor([3, 17, 3, 8])
    assert emb.fused_objective(fn, None, rep, dense=ds.pdists) is None           # no in-kernel route, no targets: caller gathers
    loss = obj(rep)
    x = emb.xs[0]
    g, = torch.autograd.grad(loss, x)
    assert bool(torch.isfinite(g).all()) and float(g[3].abs().sum()) > 0
    with pytest.raises(ValueError):
        step(indices=rep)
    # the steps above that raised must not have moved anything; a good batch still steps
    before = emb.xs[0].detach().clone()
    l1 = float(step(indices=good))
    assert abs(l1 - ref) <= 1e-5 * abs(ref) and not torch.equal(before, emb.xs[0].detach())
    # device-side garbage is clamped inside the kernels: finite numbers, nothing written outside the workspace
    junk = torch.tensor([3, 17, 2 ** 31 + 5, -7, 10 ** 9], device='cuda')
    lj = step(indices=junk)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(lj)) and bool(torch.isfinite(emb.xs[0]).all())
    lo = emb.fused_objective(fn, None, junk, dense=ds.pdists)
    gj, = torch.autograd.grad(lo, emb.xs[0])
    torch.cuda.synchronize()
    assert bool(torch.isfinite(lo)) and bool(torch.isfinite(gj).all())
