#!/usr/bin/env python3
"""Randomised cross-check of the op-level pair kernels (pdist forward + backward, all manifolds of the path)
against the plain-C fp64 checker oracle/exact.c: random sizes (n = 1, 2, tile edges, ragged), dimensions,
dtypes, spreads (close-pair series / Cayley / Jacobi regimes), squared or not, row shards.
Not collected by pytest (run by hand on a GPU box): python tests/fuzz_pdist.py [cases] [seed] [--big]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from graphembed import _backend as B  # noqa: E402
from graphembed import manifolds as M  # noqa: E402
from oracle import exact  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    big = '--big' in sys.argv
    cases = int(args[0]) if args else 200
    rng = random.Random(int(args[1]) if len(args) > 1 else 0)
    worst = {}
    for c in range(cases):
        dt = rng.choice([torch.float32, torch.float64])
        fam = rng.choice(['spd', 'spd', 'euclidean', 'lorentz', 'sphere'])
        n = rng.choice([1, 2, 3, 8, 9, 63, 64, 65, 255, 256, 257, 511, rng.randint(1, 700)])
        if big:   # tile-height switch (16-row backward tiles from 8.4 M pairs), many column blocks, ragged edges
            n = rng.choice([2047, 2049, 3001, 4095, 4097, 4160, 5003])
        torch.manual_seed(c)
        if fam == 'spd':
            d = rng.choice([2, 3, 3, 4, 5])
            man = M.SymmetricPositiveDefinite(d)
            spread = rng.choice([0.05, 0.1, 0.35, 1.0, 2.5])
            x = man.rand(n, out=torch.empty(0, dtype=torch.float64, device='cuda'), ir=spread)
            what = f'spd{d} ir={spread}'
        else:
            m = rng.randint(2 if fam != 'euclidean' else 1, 32) if rng.random() < 0.8 else rng.randint(33, 64)   # (the kernels go to 64)
            man = {'euclidean': M.Euclidean, 'lorentz': M.Lorentz, 'sphere': M.Sphere}[fam](m)
            man.use_gram = rng.random() < 0.5
            spread = rng.choice([0.01, 0.3, 1.0])
            x = man.rand(n, out=torch.empty(0, dtype=torch.float64, device='cuda'), ir=spread) if fam != 'sphere' \
                else torch.nn.functional.normalize(torch.randn(n, m, dtype=torch.float64, device='cuda'), dim=-1)
            what = f'{fam}{m} gram={man.use_gram} ir={spread}'
        xin = x.to(dt)
        x64 = xin.double().cpu().numpy()
        P = n * (n - 1) // 2
        g = torch.randn(P, dtype=dt, device='cuda')
        squared = rng.random() < 0.7
        xr = xin.clone().requires_grad_()
        world = rng.randint(1, 3)
        outs, grads = [], torch.zeros_like(xin)
        for r in range(world):
            rows = B.shard_rows(n, world, r)
            lo, hi = B.pair_offset(n, rows[0]), B.pair_offset(n, rows[1])
            d2 = man.pdist(xr, squared=squared, rows=rows)
            assert d2.numel() == hi - lo
            outs.append(d2.detach())
            if hi > lo:
                grads += torch.autograd.grad(d2, xr, g[lo:hi])[0]
        got = torch.cat(outs).double().cpu().numpy() if outs else np.zeros(0)
        if fam == 'spd':
            ref = exact.spd_pdist(x64)
            rgrad = exact.spd_pdist_grad(x64, g.double().cpu().numpy()) if squared else None
        else:
            ref = exact.vec_pdist(fam, x64)
            rgrad = exact.vec_pdist_grad(fam, x64, g.double().cpu().numpy()) if squared else None
        if not squared:
            ref = np.sqrt(ref)
        f32 = dt == torch.float32
        atol, rtol = (2e-6, 5e-5) if f32 else (1e-12, 1e-9)
        if f32 and fam == 'spd' and spread >= 1.0:   # cond(X) ~ e^(2 spread): eps * cond enters the pair matrix
            atol = 1e-4
        if not squared:   # sqrt near 0 amplifies the absolute error of d2
            atol = 2e-3 if f32 else 1e-6
        cond_ok = True
        if fam == 'sphere' and f32 and P:
            # acos'(c) = -1/sqrt(1 - c^2) with c an fp32 inner product: near-(anti)podal pairs are ill-conditioned
            # in ANY fp32 evaluation of the reference's formula (sphere.py:68-74) — skip those draws
            cc = np.clip(x64 @ x64.T, -1, 1)[np.triu_indices(n, 1)]
            cond_ok = (1 - cc * cc).min() > 1e-3
        if not cond_ok:
            atol = 2e-2
        bad = np.abs(got - ref) - (atol + rtol * np.abs(ref)) if P else np.zeros(1) - 1
        ok = bad.max() <= 0 and bool(torch.isfinite(grads).all())
        ge = 0.0
        if ok and rgrad is not None and P and cond_ok:
            scale = max(np.abs(rgrad).max(), 1e-30)
            ge = np.abs(grads.double().cpu().numpy() - rgrad).max() / scale
            # fp32 vector manifolds at tiny spread are ill-conditioned: compare where meaningful
            gtol = (5e-4 if fam == 'spd' else (5e-2 if spread <= 0.01 else 2e-3)) if f32 else 1e-7
            if fam == 'sphere' and P:   # acos'(c) amplifies the rounding of c by 1 / (1 - c^2)
                cc = np.clip(x64 @ x64.T, -1, 1)[np.triu_indices(n, 1)]
                gtol = max(gtol, 20 * (6e-8 if f32 else 1.2e-16) / max((1 - cc * cc).min(), 1e-300))
            ok = ge <= gtol
        key = (fam, 'f32' if f32 else 'f64')
        worst[key] = max(worst.get(key, 0.0), ge)
        if not ok:
            print(f'FAIL case {c}: {what} n={n} {dt} squared={squared} world={world} '
                  f'd2 excess {bad.max():.2e} grad err {ge:.2e}')
            sys.exit(1)
    print(f'{cases} cases ok; worst grad rel err: ' + ', '.join(f'{k[0]}/{k[1]} {v:.1e}' for k, v in sorted(worst.items())))


if __name__ == '__main__':
    main()
