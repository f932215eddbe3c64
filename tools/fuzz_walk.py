#!/usr/bin/env python3
"""Randomised check of the resident-grid pair kernels' LAUNCH arithmetic (round 5: shares cut on the host, a share's first
column block in closed form, block-entry costs — csrc/spd_ws.hpp WalkShares / ColWalk::find_fast / ColWalk::enter): random n,
random contiguous row ranges, SPD(2..4), Lorentz / sphere / Euclidean, fp32 / fp64 —
  * the gradient of arbitrary row shards, summed, equals the unsharded gradient (every pair visited exactly once, whatever
    the cut), and the pair vector of the shards concatenates to the full one;
  * distances and gradients agree with the fp64 C checker (oracle/exact.c) for n <= 260;
  * the fused loss of shards sums to the loss of the whole.
MM_SPD_BWD_CROSS / MM_VEC_BWD_CROSS in the environment change the block-entry cost (0 = equal shares).
Usage: python tools/fuzz_walk.py [cases] [seed]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'matrix-manifolds_amd'))
import torch  # noqa: E402
from graphembed import manifolds as M  # noqa: E402
from oracle import exact  # noqa: E402


def cuts(n, rng):
    k = rng.choice([1, 2, 3, 5, 8])
    pts = sorted(set([0, n] + [rng.randint(0, n) for _ in range(k - 1)]))
    return list(zip(pts[:-1], pts[1:]))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for c in range(cases):
        dt = rng.choice([torch.float32, torch.float64])
        n = rng.choice([2, 3, 63, 64, 65, 127, 128, 129, 130, 257, rng.randint(2, 700), rng.randint(2, 700), rng.randint(700, 2600)])
        kind = rng.choice(['spd2', 'spd3', 'spd3', 'spd4', 'lorentz', 'sphere', 'euclidean'])
        torch.manual_seed(rng.randint(0, 10 ** 6))
        if kind.startswith('spd'):
            d = int(kind[3])
            man = M.SymmetricPositiveDefinite(d)
            a = torch.randn(n, d, d, dtype=torch.float64) * rng.choice([0.05, 0.2, 0.5])
            x = torch.linalg.matrix_exp(0.5 * (a + a.transpose(1, 2))).to(dt).cuda()
        else:
            m = rng.randint(2, 14)
            man = {'lorentz': M.Lorentz, 'sphere': M.Sphere, 'euclidean': M.Euclidean}[kind](m)
            x = man.rand(n, out=torch.empty(0, device='cuda', dtype=dt), ir=0.5) if kind != 'euclidean' else torch.randn(n, m, dtype=dt, device='cuda')
        npairs = n * (n - 1) // 2
        g = torch.randn(npairs, dtype=dt, device='cuda')
        xr = x.clone().requires_grad_()
        d2 = man.pdist(xr, squared=True)
        grad, = torch.autograd.grad(d2, xr, g)
        tol = 3e-5 if dt == torch.float32 else 1e-10
        scale = float(grad.abs().max()) + 1e-30
        # shards
        parts, gsum = [], torch.zeros_like(grad)
        for (rb, re) in cuts(n, rng):
            xs = x.clone().requires_grad_()
            lo, hi = rb * (2 * n - rb - 1) // 2, re * (2 * n - re - 1) // 2
            dd = man.pdist(xs, squared=True, rows=(rb, re))
            assert dd.numel() == hi - lo, (n, rb, re, dd.numel())
            parts.append(dd.detach())
            if hi > lo:
                gs, = torch.autograd.grad(dd, xs, g[lo:hi])
                gsum += gs
        cat = torch.cat(parts) if parts else d2.detach()[:0]
        e1 = float((cat - d2.detach()).abs().max()) if npairs else 0.0
        e2 = float((gsum - grad).abs().max()) / scale
        assert e1 == 0.0, (c, kind, n, dt, 'shard distances differ', e1)
        assert e2 <= tol, (c, kind, n, dt, 'sum of shard gradients', e2)
        # the C checker
        e3 = 0.0
        if n <= 260 and npairs:
            x64 = x.double().cpu().numpy()
            if kind.startswith('spd'):
                rd, rg = exact.spd_pdist(x64), exact.spd_pdist_grad(x64, g.double().cpu().numpy())
            else:
                rd, rg = exact.vec_pdist(kind, x64), exact.vec_pdist_grad(kind, x64, g.double().cpu().numpy())
            ed = np.abs(d2.detach().double().cpu().numpy() - rd) - (1e-6 + 3e-5 * np.abs(rd) if dt == torch.float32 else 1e-12 + 1e-10 * np.abs(rd))
            assert ed.max() <= 0, (c, kind, n, dt, 'd2 against the checker', ed.max())
            e3 = float(np.abs(grad.double().cpu().numpy() - rg).max() / (np.abs(rg).max() + 1e-30))
            assert e3 <= (1e-4 if dt == torch.float32 else 1e-9), (c, kind, n, dt, 'gradient against the checker', e3)
        worst = max(worst, e2, e3 if dt == torch.float64 else 0.0)
    print(f'fuzz_walk: {cases} cases ok (worst relative gradient deviation between shard sums and the whole {worst:.2e})')


if __name__ == '__main__':
    main()
