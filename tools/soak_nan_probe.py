"""Where does the fp32 config-4 soak go non-finite?  Steps the product embedding with Adam through (a) the one-call step and (b) the
eager per-factor path (pair_kernel off: each factor's own pdist kernels + autograd + the optimizers), reporting per 100 steps the loss,
the largest |coordinate| of every factor and the Lorentz constraint residual max |<x,x>_L + 1|."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'matrix-manifolds_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402
from graphembed import manifolds as M  # noqa: E402
from graphembed.modules import ManifoldEmbedding  # noqa: E402
from graphembed.native_step import NativeTrainStep  # noqa: E402
from graphembed.objectives import StressLoss  # noqa: E402
from graphembed.optim import RiemannianAdam  # noqa: E402
from train_soak import tree_distances  # noqa: E402

n, steps = 1025, 3000
target = tree_distances(n, torch.Generator().manual_seed(0)).float().cuda()
for mode in ('native', 'eager_perfactor'):
    torch.manual_seed(1)
    with torch.device('cuda'):
        emb = ManifoldEmbedding(n, [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)])
    fn = StressLoss()
    opts = [RiemannianAdam(list(emb.xs), lr=1e-2, exact=True, max_grad_norm=20), RiemannianAdam(list(emb.scales), lr=1e-3)]
    if mode == 'native':
        step = NativeTrainStep(emb, fn, target, opts)
    else:
        emb.pair_kernel = False
    for k in range(steps):
        if mode == 'native':
            loss = step(epoch=3, alpha=1.0)
        else:
            loss = emb.fused_objective(fn, target, None, epoch=3, alpha=1.0)
            for o in opts:
                o.zero_grad(set_to_none=True)
            loss.backward()
            for o in opts:
                o.step()
        if k % 100 == 0 or k == steps - 1 or not bool(torch.isfinite(loss)):
            xl = emb.xs[0].detach()
            res = (-(xl[:, 0] ** 2) + (xl[:, 1:] ** 2).sum(-1) + 1).abs().max().item()
            print(mode, k, f'loss {float(loss):.5g}', 'max|x|', [f'{x.detach().abs().max().item():.3g}' for x in emb.xs],
                  f'Lorentz residual {res:.3g}', 'scales', [f'{float(s):.3g}' for s in emb.scales], flush=True)
            if not bool(torch.isfinite(loss)):
                break
