"""Golden vectors of the node-minibatch objective from the REAL reference (development container only).
    PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0 python tests/golden/gen_golden_minibatch.py
The body of the training loop for one sampled node set (graphembed/train.py:206-213, modules.py:84-105,
data/dataset.py:19-27):  loss = objective_fn(dataset[idx], embedding.compute_dists(idx), epoch, alpha), with its gradients
w.r.t. every factor's points (full shape: zero rows outside the batch) and every scale.  Product H^6 x S^6 x SPD(2) and the
single factors SPD(3), Lorentz(11); fp32 + fp64; StressLoss and QuotientLoss.  Output tests/golden/minibatch.npz."""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from graphembed import manifolds as RM  # noqa: E402
from graphembed.data import GraphDataset  # noqa: E402
from graphembed.modules import ManifoldEmbedding  # noqa: E402
from graphembed.objectives import QuotientLoss, StressLoss  # noqa: E402
from gen_golden import np_, DT  # noqa: E402

CASES = {
    'product': lambda: [RM.Lorentz(6), RM.Sphere(6), RM.SymmetricPositiveDefinite(2)],
    'spd3': lambda: [RM.SymmetricPositiveDefinite(3)],
    'lorentz11': lambda: [RM.Lorentz(11)],
}
# round 4 (`--wide`, output minibatch_wide.npz): the single factors whose minibatches run inside their OWN pair kernels
# (mm_spd_pdist_loss_subset / mm_vec_pdist_loss_subset) — BASELINE config 5's SPD(4), a Jacobi-path SPD(6), vectors wider than 16
WIDE = {
    'spd4': lambda: [RM.SymmetricPositiveDefinite(4)],
    'spd6': lambda: [RM.SymmetricPositiveDefinite(6)],
    'lorentz24': lambda: [RM.Lorentz(24)],
    'sphere20': lambda: [RM.Sphere(20)],
    'euclidean40': lambda: [RM.Euclidean(40)],
}


def main():
    out = {}
    n, bs = 61, 23
    wide = '--wide' in sys.argv
    for name, mans in (WIDE if wide else CASES).items():
        for dname in DT:
            torch.set_default_dtype(DT[dname])
            torch.manual_seed(zlib.crc32(repr((name, dname, 'minibatch')).encode()) % (2**31))
            emb = ManifoldEmbedding(n, mans())
            with torch.no_grad():
                for k, s in enumerate(emb.scales):
                    s.fill_(0.3 + 0.3 * k)
                emb.perturb(0.3)
            graph_d = torch.rand(n * (n - 1) // 2) * 3 + 0.5          # "graph distances" of all pairs
            ds = GraphDataset(graph_d.clone())
            idx = torch.randperm(n)[:bs]
            base = f'{name}/{dname}'
            out[f'{base}/graph_d'] = np_(graph_d)
            out[f'{base}/idx'] = np_(idx)
            out[f'{base}/scales'] = np.array([s.item() for s in emb.scales])
            for k, x in enumerate(emb.xs):
                out[f'{base}/x_{k}'] = np_(x.data)
            for lname, fn, kw in (('stress', StressLoss(), {}), ('quotient', QuotientLoss(), dict(epoch=2, alpha=1.3))):
                md = emb.compute_dists(idx)
                loss = fn(ds[idx], md, **kw)
                grads = torch.autograd.grad(loss, list(emb.xs) + list(emb.scales))
                assert torch.isfinite(loss) and all(torch.isfinite(g).all() for g in grads)
                out[f'{base}/{lname}/loss'] = np_(loss)
                for k in range(len(emb.xs)):
                    out[f'{base}/{lname}/grad_x_{k}'] = np_(grads[k])
                    out[f'{base}/{lname}/grad_s_{k}'] = np_(grads[len(emb.xs) + k])
    np.savez_compressed(os.path.join(HERE, 'minibatch_wide.npz' if wide else 'minibatch.npz'), **out)
    print(len(out), 'arrays')
    torch.set_default_dtype(torch.float32)


if __name__ == '__main__':
    main()
