"""Generate golden vectors by running the REAL reference (development container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Imports graphembed from /root/reference behind ``ref_shim`` and records, for
every manifold on the hot path, inputs and the reference's outputs:
``pdist`` (squared and not), autograd gradients, the optimizer-side maps
(``egrad2rgrad / norm / exp / retr / projx / log / transp``), RiemannianSGD
steps, the product embedding's ``compute_dists`` and short full-batch training
traces on the tree40 graph.  Outputs: ``tests/golden/*.npz`` (data only — no
reference source travels).
"""
import gzip
import itertools
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
import graphembed  # noqa: E402  (the reference)
from graphembed import manifolds as RM  # noqa: E402
from graphembed.modules import ManifoldEmbedding, ManifoldParameter  # noqa: E402
from graphembed.objectives import QuotientLoss, StressLoss  # noqa: E402
from graphembed.optim import RiemannianSGD  # noqa: E402
from graphembed.data import GraphDataset  # noqa: E402

DT = {'f32': torch.float32, 'f64': torch.float64}


def np_(t):
    return t.detach().cpu().numpy()


def seed_of(*parts):
    """A seed that is the same in every process (the built-in hash() of a tuple of strings is salted per process)."""
    return zlib.crc32(repr(parts).encode()) % (2**31)


def rand_spd(n, d):  # tests/conftest.py:36-43 of the reference
    x = torch.rand(n, d, d)
    return (x @ x.transpose(1, 2)).add_(torch.eye(d))


MANIFOLDS = {
    # key: (constructor, point inits {name: fn(man, n)})
    'spd2': (lambda: RM.SymmetricPositiveDefinite(2), None),
    'spd3': (lambda: RM.SymmetricPositiveDefinite(3), None),
    'spd4': (lambda: RM.SymmetricPositiveDefinite(4), None),
    'spd5': (lambda: RM.SymmetricPositiveDefinite(5), None),
    'spd6': (lambda: RM.SymmetricPositiveDefinite(6), None),
    'spd7': (lambda: RM.SymmetricPositiveDefinite(7), None),
    'spd8': (lambda: RM.SymmetricPositiveDefinite(8), None),
    'spd9': (lambda: RM.SymmetricPositiveDefinite(9), None),
    'lorentz11': (lambda: RM.Lorentz(11), None),
    'lorentz6': (lambda: RM.Lorentz(6), None),
    'lorentz3': (lambda: RM.Lorentz(3), None),
    'lorentz48': (lambda: RM.Lorentz(48), None),
    'sphere64': (lambda: RM.Sphere(64), None),
    'euclidean40': (lambda: RM.Euclidean(40), None),
    'sphere6': (lambda: RM.Sphere(6), None),
    'euclidean10': (lambda: RM.Euclidean(10), None),
    'grassmann52': (lambda: RM.Grassmann(5, 2), None),
    'grassmann63': (lambda: RM.Grassmann(6, 3), None),
    'stiefel52': (lambda: RM.Stiefel(5, 2), None),
}


def make_points(key, man, n, init):
    if init == 'rand':  # the reference's own initialisation
        return man.rand(n)
    # "wide": trained-like / well-separated points
    if key.startswith('spd'):
        return rand_spd(n, man.n)
    if key.startswith('lorentz'):
        return man.rand(n, ir=0.5)
    if key.startswith('sphere') or key.startswith('grassmann') or key.startswith('stiefel'):
        return man.rand_uniform(n)
    return torch.randn(n, *man.shape)


def tangent_like(key, man, x, scale):
    u = torch.randn_like(x) * scale
    if key.startswith('spd'):
        return 0.5 * (u + u.transpose(-2, -1))
    return man.proju(x, u)


def record_manifold(key, n, out):
    ctor, _ = MANIFOLDS[key]
    for dname, init in itertools.product(DT, ['rand', 'wide']):
        torch.set_default_dtype(DT[dname])
        torch.manual_seed(seed_of(key, dname, init, n))
        man = ctor()
        x = make_points(key, man, n, init).detach().clone()
        tag = f'{dname}/{init}/n{n}'
        out[f'{tag}/x'] = np_(x)
        P = n * (n - 1) // 2
        has_dist = not key.startswith('stiefel')
        if has_dist:
            g = torch.randn(P)
            out[f'{tag}/g'] = np_(g)
            xr = x.clone().requires_grad_()
            d2 = man.pdist(xr, squared=True)
            out[f'{tag}/d2'] = np_(d2)
            out[f'{tag}/grad_d2'] = np_(torch.autograd.grad((d2 * g).sum(), xr)[0])
            xr = x.clone().requires_grad_()
            d1 = man.pdist(xr, squared=False)
            out[f'{tag}/d1'] = np_(d1)
            out[f'{tag}/grad_d1'] = np_(torch.autograd.grad((d1 * g).sum(), xr)[0])
            # dist on explicit (x, y) pairs, y = reversed x
            y = x.flip(0).clone()
            out[f'{tag}/dist_xy'] = np_(man.dist(x, y, squared=True))
            egrad = torch.from_numpy(out[f'{tag}/grad_d2']).clone()
        else:
            egrad = torch.randn_like(x)
            out[f'{tag}/egrad_in'] = np_(egrad)
        with torch.no_grad():
            rg = man.egrad2rgrad(x.clone(), egrad.clone())
            out[f'{tag}/rgrad'] = np_(rg)
            out[f'{tag}/rgrad_norm'] = np_(man.norm(x, rg, keepdim=True))
            u = tangent_like(key, man, x, 0.3)
            out[f'{tag}/u'] = np_(u)
            out[f'{tag}/proju'] = np_(man.proju(x.clone(), u.clone()))
            out[f'{tag}/norm_u'] = np_(man.norm(x, man.proju(x.clone(), u.clone()), keepdim=True))
            pu = man.proju(x.clone(), u.clone())
            if not key.startswith('stiefel'):
                out[f'{tag}/exp'] = np_(man.exp(x.clone(), pu.clone()))
                out[f'{tag}/log'] = np_(man.log(x.clone(), x.flip(0).clone()))
            out[f'{tag}/retr'] = np_(man.retr(x.clone(), pu.clone()))
            y = man.retr(x.clone(), pu.clone())
            out[f'{tag}/transp'] = np_(man.transp(x.clone(), y, pu.clone()))
            xp = x.clone() + 0.05 * torch.randn_like(x)  # off-manifold point
            out[f'{tag}/projx_in'] = np_(xp)
            if key.startswith('stiefel'):
                out[f'{tag}/projx'] = np_(man._orthonormalize(xp.clone()))
                out[f'{tag}/retr_qr'] = np_(man.retr_qr_(x.clone(), pu.clone()))
            else:
                out[f'{tag}/projx'] = np_(man.projx(xp.clone()))
            if key.startswith('grassmann'):
                out[f'{tag}/retr_qr'] = np_(man.retr_qr_(x.clone(), pu.clone()))


def record_rsgd(key, n, out):
    """Two consecutive RiemannianSGD steps with fixed Euclidean gradients."""
    ctor, _ = MANIFOLDS[key]
    for dname in DT:
        torch.set_default_dtype(DT[dname])
        torch.manual_seed(seed_of(key, dname, 'rsgd'))
        man = ctor()
        x0 = make_points(key, man, n, 'wide').detach().clone()
        # (SPD(6..9): smaller Euclidean gradients — the Riemannian gradient X G X grows with the dimension, and the
        # reference's unclipped second-order retraction leaves the cone for steps that large)
        gscale = {'spd5': 1.0, 'spd6': 0.3, 'spd7': 0.3, 'spd8': 0.3, 'spd9': 0.3}.get(key, 3.0)
        g1 = torch.randn_like(x0) * gscale
        g2 = torch.randn_like(x0) * gscale
        if key.startswith('spd'):
            g1 = g1 + g1.transpose(-2, -1)
            g2 = g2 + g2.transpose(-2, -1)
        base = f'{dname}/rsgd'
        out[f'{base}/x0'], out[f'{base}/g1'], out[f'{base}/g2'] = np_(x0), np_(g1), np_(g2)
        for exact, mgn, mom in itertools.product([False, True], [None, 2.0], [0.0, 0.9]):
            if exact and key.startswith('stiefel'):
                continue
            p = ManifoldParameter(x0.clone(), manifold=man)
            opt = RiemannianSGD([p], lr=0.05, momentum=mom, dampening=0.1 if mom else 0,
                                max_grad_norm=mgn, exact=exact)
            tag = f'{base}/exact{int(exact)}_clip{0 if mgn is None else 1}_mom{int(mom > 0)}'
            p.grad = g1.clone()
            opt.step()
            out[f'{tag}/x1'] = np_(p.data)
            p.grad = g2.clone()
            opt.step()
            out[f'{tag}/x2'] = np_(p.data)
            assert np.isfinite(out[f'{tag}/x2']).all(), f'{key} {tag}: the reference itself left the manifold (reduce gscale)'
            if mom > 0:
                out[f'{tag}/buf2'] = np_(opt.state[p]['momentum_buffer'])


def load_tree40_targets():
    import networkx as nx
    from scipy.sparse.csgraph import shortest_path
    from scipy.spatial.distance import squareform
    with gzip.open('/root/reference/data/tree40.edges.gz', 'rt') as f:
        g = nx.parse_edgelist((l for l in f if l.strip() and not l.startswith('#')),
                              nodetype=int, data=False)
    g = nx.convert_node_labels_to_integers(g, ordering='sorted')
    a = nx.to_scipy_sparse_array(g, nodelist=range(len(g)))
    d = shortest_path(a, unweighted=True, directed=False)
    return torch.from_numpy(squareform(d, checks=False)).to(torch.get_default_dtype())


def record_training(out):
    """Full-batch RSGD traces on tree40 built from the reference's components
    (GraphDataset -> ManifoldEmbedding.compute_dists -> loss -> RiemannianSGD)."""
    cases = {
        'euclidean10': lambda: [RM.Euclidean(10)],
        'lorentz11': lambda: [RM.Lorentz(11)],
        'spd3': lambda: [RM.SymmetricPositiveDefinite(3)],
        'product': lambda: [RM.Lorentz(6), RM.Sphere(6), RM.SymmetricPositiveDefinite(2)],
    }
    torch.set_default_dtype(torch.float64)
    gp = load_tree40_targets()
    out['tree40/gpdists'] = np_(gp)
    ds = GraphDataset(gp.clone())
    target = ds[None]
    out['tree40/target'] = np_(target)
    for name, mk in cases.items():
        for loss_name in ['stress', 'quotient']:
            torch.manual_seed(7)
            emb = ManifoldEmbedding(40, mk())
            for k, x in enumerate(emb.xs):
                out[f'tree40/{name}/{loss_name}/x0_{k}'] = np_(x.data)
            opt = RiemannianSGD(list(emb.xs), lr=0.01, exact=True, max_grad_norm=20)
            opt_s = RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)
            fn = StressLoss() if loss_name == 'stress' else QuotientLoss()
            losses = []
            for epoch in range(20):
                md = emb.compute_dists(None)
                loss = fn(target, md, epoch=epoch, alpha=1.0)
                opt.zero_grad()
                opt_s.zero_grad()
                loss.backward()
                opt.step()
                opt_s.step()
                losses.append(loss.item())
            out[f'tree40/{name}/{loss_name}/losses'] = np.array(losses)
            for k, x in enumerate(emb.xs):
                out[f'tree40/{name}/{loss_name}/x20_{k}'] = np_(x.data)
            out[f'tree40/{name}/{loss_name}/scales20'] = np.array(
                [s.item() for s in emb.scales])


def record_product(out):
    for dname in DT:
        torch.set_default_dtype(DT[dname])
        torch.manual_seed(11)
        n = 33
        emb = ManifoldEmbedding(n, [RM.Lorentz(6), RM.Sphere(6), RM.SymmetricPositiveDefinite(2)])
        with torch.no_grad():
            emb.scales[0].fill_(0.3)
            emb.scales[1].fill_(0.5)
            emb.scales[2].fill_(0.9)
            emb.perturb(0.4)
        g = torch.randn(n * (n - 1) // 2)
        idx = torch.randperm(n)[:17]
        for tag, ii in [('full', None), ('batch', idx)]:
            md = emb.compute_dists(ii)
            gg = g[:md.numel()]
            grads = torch.autograd.grad((md * gg).sum(), list(emb.xs) + list(emb.scales))
            base = f'{dname}/product/{tag}'
            out[f'{base}/d2'] = np_(md)
            out[f'{base}/g'] = np_(gg)
            for k in range(3):
                out[f'{base}/x_{k}'] = np_(emb.xs[k].data)
                out[f'{base}/grad_x_{k}'] = np_(grads[k])
                out[f'{base}/grad_s_{k}'] = np_(grads[3 + k])
            out[f'{base}/scales'] = np.array([s.item() for s in emb.scales])
        out[f'{dname}/product/idx'] = np_(idx)
        gd = torch.rand(n * (n - 1) // 2) + 0.01
        md = emb.compute_dists(None).detach()
        out[f'{dname}/loss/gd'] = np_(gd)
        out[f'{dname}/loss/md'] = np_(md)
        out[f'{dname}/loss/stress'] = np_(StressLoss()(gd, md))
        out[f'{dname}/loss/quotient'] = np_(QuotientLoss()(gd, md, epoch=3, alpha=1.7))


def main():
    """`gen_golden.py` regenerates everything; `gen_golden.py key [key ...]` only the named manifold files (seeds are per
    key, so a partial run reproduces exactly what a full run writes for those keys)."""
    torch.set_num_threads(4)
    only = set(sys.argv[1:])
    unknown = only - set(MANIFOLDS) - {'callers'}
    if unknown:
        raise SystemExit(f'unknown keys {sorted(unknown)}')
    for key in MANIFOLDS:
        if only and key not in only:
            continue
        out = {}
        record_manifold(key, 33, out)
        if key in ('spd3', 'lorentz11'):
            record_manifold(key, 96, out)
        record_rsgd(key, 17, out)
        np.savez_compressed(os.path.join(HERE, f'{key}.npz'), **out)
        print(key, len(out), 'arrays')
    if not only or 'callers' in only:
        out = {}
        record_product(out)
        record_training(out)
        np.savez_compressed(os.path.join(HERE, 'callers.npz'), **out)
        print('callers', len(out), 'arrays')
    torch.set_default_dtype(torch.float32)


if __name__ == '__main__':
    main()
