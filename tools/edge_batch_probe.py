import sys, torch
sys.path.insert(0, '/root/repo/matrix-manifolds_amd')
from graphembed import manifolds as M
from graphembed.modules import ManifoldEmbedding
from graphembed.native_step import NativeTrainStep
from graphembed.objectives import StressLoss, QuotientLoss
from graphembed.optim import RiemannianSGD
n = 37
for mans in ([M.SymmetricPositiveDefinite(4)], [M.SymmetricPositiveDefinite(3)], [M.Lorentz(24)], [M.Euclidean(5)],
             [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)]):
    for fn in (StressLoss(), QuotientLoss()):
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, mans)
        dense = torch.rand(n, n, device='cuda'); dense = (dense + dense.t()).contiguous(); dense.fill_diagonal_(0)
        for bs in (0, 1, 2):
            idx = torch.randperm(n, device='cuda')[:bs]
            try:
                loss = emb.fused_objective(fn, None, idx, dense=dense, epoch=1, alpha=1.0)
                if loss is None:
                    print([str(m) for m in mans], type(fn).__name__, 'bs', bs, '-> no fused route (None)')
                    continue
                gs = torch.autograd.grad(loss, list(emb.xs) + list(emb.scales), allow_unused=True)
                ok = all(g is None or bool(torch.isfinite(g).all()) for g in gs)
                nz = max(float(g.abs().max()) for g in gs if g is not None)
                print([str(m)[:12] for m in mans], type(fn).__name__, 'bs', bs, 'loss', float(loss), 'finite', ok, 'max|grad|', nz)
            except Exception as e:
                print([str(m)[:12] for m in mans], type(fn).__name__, 'bs', bs, 'EXC', type(e).__name__, str(e)[:120])
        if len(mans) == 1:
            opts = [RiemannianSGD(list(emb.xs), lr=1e-3), RiemannianSGD(list(emb.scales), lr=1e-4)]
            step = NativeTrainStep(emb, fn, None, opts, dense=dense)
            for bs in (0, 1, 2):
                idx = torch.randperm(n, device='cuda')[:bs]
                try:
                    l = step(indices=idx, epoch=1, alpha=1.0)
                    print('   native step bs', bs, 'loss', float(l), 'finite points', bool(torch.isfinite(emb.xs[0]).all()))
                except Exception as e:
                    print('   native step bs', bs, 'EXC', type(e).__name__, str(e)[:120])
