#!/usr/bin/env python3
"""Soak: full-batch training through the one-call step (mm_train_step_run replayed eagerly) for thousands of steps on synthetic
graph distances, fp32 next to fp64 from the same start — the embedding walks through the kernels' regimes (close-pair series ->
recentred series / Cayley -> Jacobi) at full scale.  The points are re-projected every 20 steps, as the reference's engine does (train.py:184-187, example_config.yaml:37) — without
it fp32 Lorentz points leave the hyperboloid within ~1000 Adam steps in BOTH implementations (tools/soak_nan_probe.py).
Checks: every loss finite, the loss goes down, fp32 follows fp64.
    python tools/train_soak.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'matrix-manifolds_amd'))
import torch  # noqa: E402
from graphembed import manifolds as M  # noqa: E402
from graphembed.modules import ManifoldEmbedding  # noqa: E402
from graphembed.native_step import NativeTrainStep  # noqa: E402
from graphembed.objectives import QuotientLoss, StressLoss  # noqa: E402
from graphembed.optim import RiemannianAdam, RiemannianSGD  # noqa: E402


def tree_distances(n, gen):
    """Squared hop distances of a random tree (parents drawn uniformly among earlier nodes), normalised to max 1."""
    parent = torch.zeros(n, dtype=torch.int64)
    depth = torch.zeros(n, dtype=torch.int64)
    for v in range(1, n):
        parent[v] = int(torch.randint(0, v, (1, ), generator=gen))
        depth[v] = depth[parent[v]] + 1
    anc = [set() for _ in range(n)]
    d = torch.zeros(n, n)
    paths = []
    for v in range(n):
        p, path = v, []
        while True:
            path.append(p)
            if p == 0:
                break
            p = int(parent[p])
        paths.append(path)
    for a in range(n):
        pa = {node: k for k, node in enumerate(paths[a])}
        for b in range(a + 1, n):
            for k, node in enumerate(paths[b]):
                if node in pa:
                    d[a, b] = d[b, a] = pa[node] + k
                    break
    d = d / d.max()
    iu = torch.triu_indices(n, n, 1)
    return (d[iu[0], iu[1]] ** 2).clamp_min(1e-4)


def run(name, mans_of, n, loss_name, opt_name, steps, lr):
    gen = torch.Generator().manual_seed(0)
    target64 = tree_distances(n, gen).double()
    out = {}
    for dt in (torch.float64, torch.float32):
        torch.manual_seed(1)
        torch.set_default_dtype(dt)
        try:
            with torch.device('cuda'):
                emb = ManifoldEmbedding(n, mans_of())
        finally:
            torch.set_default_dtype(torch.float32)
        fn = StressLoss() if loss_name == 'stress' else QuotientLoss()
        if opt_name == 'rsgd':
            opts = [RiemannianSGD(list(emb.xs), lr=lr, exact=True, max_grad_norm=20), RiemannianSGD(list(emb.scales), lr=lr * 0.1, max_grad_norm=500)]
        else:
            opts = [RiemannianAdam(list(emb.xs), lr=lr, exact=True, max_grad_norm=20), RiemannianAdam(list(emb.scales), lr=lr * 0.1)]
        step = NativeTrainStep(emb, fn, target64.to(dt).cuda(), opts)
        losses = []
        t0 = time.perf_counter()
        for k in range(steps):
            if k and k % 20 == 0:     # stabilize_every_epochs = 20 (example_config.yaml:37; run_grid.py:134 uses 5): train.py:184-187
                with torch.no_grad():
                    emb.stabilize()
            l = step(epoch=3, alpha=1.0)      # (a fixed epoch: the quotient loss's eps = 1 / (epoch + 1) changes the objective itself)
            if k % max(1, steps // 20) == 0 or k == steps - 1:
                losses.append(float(l))
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
        finite = all(v == v and abs(v) < 1e30 for v in losses) and all(bool(torch.isfinite(x).all()) for x in emb.xs)
        out[dt] = (losses, finite, dtm)
    l64, l32 = out[torch.float64][0], out[torch.float32][0]
    dev = max(abs(a - b) / max(abs(a), 1e-30) for a, b in zip(l64, l32))
    ok = out[torch.float64][1] and out[torch.float32][1] and l64[-1] < l64[0] and l32[-1] < l32[0]
    print(f'{name}: n={n} {loss_name}/{opt_name} {steps} steps: fp64 loss {l64[0]:.4g} -> {l64[-1]:.4g}, fp32 {l32[0]:.4g} -> {l32[-1]:.4g}; '
          f'largest fp32/fp64 loss deviation at the sampled steps {dev:.2e}; {out[torch.float32][2] / steps * 1e6:.0f} us/step fp32, '
          f'{out[torch.float64][2] / steps * 1e6:.0f} fp64; {"ok" if ok else "FAIL"}', flush=True)
    return ok


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    ok = True
    ok &= run('config 3 SPD(3)', lambda: [M.SymmetricPositiveDefinite(3)], 1500, 'stress', 'rsgd', steps, 5e-3)
    ok &= run('config 5 SPD(4)', lambda: [M.SymmetricPositiveDefinite(4)], 1200, 'quotient', 'adam', steps, 1e-2)
    ok &= run('SPD(3), Adam', lambda: [M.SymmetricPositiveDefinite(3)], 1500, 'quotient', 'adam', steps, 2e-2)
    ok &= run('SPD(6), Adam (matrix series -> recentred -> Jacobi)', lambda: [M.SymmetricPositiveDefinite(6)], 600, 'quotient', 'adam', steps, 2e-2)
    ok &= run('config 2 Lorentz(11)', lambda: [M.Lorentz(11)], 1500, 'stress', 'rsgd', steps, 5e-4)
    ok &= run('config 2 Lorentz(11), Adam', lambda: [M.Lorentz(11)], 1500, 'quotient', 'adam', steps, 1e-2)
    ok &= run('config 4 H6 x S6 x SPD(2)', lambda: [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, 'stress', 'adam', steps, 1e-2)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
