#!/usr/bin/env python3
"""Randomised cross-check of the remaining pair kernels against the reference-faithful torch port
oracle/ref_port.py (autograd on CPU, fp64 inputs): Stein divergence forward + backward (SPD(2..5)), Grassmann
principal-angle pdist forward + backward — random sizes incl. n = 1, 2 and tile edges, row shards.
Not collected by pytest (run by hand on a GPU box): python tests/fuzz_misc.py [cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from graphembed import _backend as B  # noqa: E402
from graphembed import manifolds as M  # noqa: E402
from oracle import ref_port as rp  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = {}
    for c in range(cases):
        torch.manual_seed(c)
        dt = rng.choice([torch.float32, torch.float64])
        n = rng.choice([1, 2, 3, 63, 64, 65, 129, rng.randint(1, 300)])
        fam = rng.choice(['stein', 'grassmann'])
        p = 0
        squared = rng.random() < 0.7
        if fam == 'stein':
            d = rng.choice([2, 3, 4, 5])
            man, ref = M.SymmetricPositiveDefinite(d, use_stein_div=True), rp.SPD(d)
            x64 = ref.rand(n, ir=rng.choice([0.1, 0.5, 1.5]), dtype=torch.float64)
            ref_pd = lambda x: ref.stein_pdiv(x, squared=squared)  # noqa: E731
            what = f'stein{d}'
        else:
            N, p = rng.choice([(3, 1), (4, 1), (4, 2), (5, 2), (6, 3), (9, 4)])
            man, ref = M.Grassmann(N, p), rp.make('grassmann', N, p)
            x64 = torch.linalg.qr(torch.randn(n, N, p, dtype=torch.float64))[0]
            ref_pd = lambda x: ref.pdist(x, squared=squared)  # noqa: E731
            what = f'gr({N},{p})'
        xin = x64.to(dt)
        P = n * (n - 1) // 2
        g = torch.randn(P, dtype=torch.float64)
        xr = xin.double().clone().requires_grad_()
        dref = ref_pd(xr)
        gref = torch.autograd.grad(dref, xr, g)[0] if P else torch.zeros_like(xr)
        if fam == 'stein':
            gref = 0.5 * (gref + gref.transpose(-2, -1))
        xg = xin.cuda().requires_grad_()
        world = rng.randint(1, 3)
        outs, grads = [], torch.zeros_like(xg)
        for r in range(world):
            rows = B.shard_rows(n, world, r)
            lo, hi = B.pair_offset(n, rows[0]), B.pair_offset(n, rows[1])
            d2 = man.pdist(xg, squared=squared, rows=rows)
            outs.append(d2.detach())
            if hi > lo:
                grads = grads + torch.autograd.grad(d2, xg, g[lo:hi].to(dt).cuda())[0]
        got = torch.cat(outs).double().cpu()
        f32 = dt == torch.float32
        if not squared and P and float(dref.detach().min()) < (1e-2 if f32 else 1e-6):
            continue   # sqrt at ~0: ill-conditioned draw (coincident points)
        if fam == 'grassmann' and P:   # acos' near sigma = 1 (nearly coincident subspaces): ill-conditioned
            if float(dref.detach().min()) < (1e-2 if f32 else 1e-5):
                continue
        de = float((got - dref.detach()).abs().max() / dref.detach().abs().max().clamp(min=1e-30)) if P else 0.0
        ge = float((grads.double().cpu() - gref).abs().max() / gref.abs().max().clamp(min=1e-30)) if P else 0.0
        key = (fam, 'f32' if f32 else 'f64')
        worst[key] = max(worst.get(key, 0.0), de, ge)
        tol = 2e-3 if f32 else 1e-7
        # fp32 acos' of singular values; p = 2 is the reference's closed form (fast.py:138-159), whose
        # (a1 - a2) / R difference quotient is ill-conditioned in fp32 when the two singular values nearly coincide
        gtol = (0.3 if p == 2 else 5e-2) if (fam == 'grassmann' and f32) else tol
        if not (de <= tol and ge <= gtol and bool(torch.isfinite(grads).all())):
            print(f'FAIL case {c}: {what} n={n} {dt} squared={squared} world={world} d err {de:.2e} grad err {ge:.2e} '
                  f'max|grad| {float(grads.abs().max()):.3e} max|ref grad| {float(gref.abs().max()):.3e} '
                  f'min d {float(dref.detach().min()):.3e} bad entries {int((grads.abs() > 1e6).sum())}')
            if fam == 'grassmann' and P:   # which pair: cosines of the principal angles of the worst node's pairs (fp64)
                err = (grads.double().cpu() - gref).abs().reshape(n, -1).max(1).values
                i = int(err.argmax())
                sv = torch.linalg.svdvals(x64[i].transpose(-1, -2).unsqueeze(0) @ x64)   # [n, p]
                sv[i] = 0.5
                print(f'  worst node {i}: gradient error {float(err[i]):.3e}; over its pairs: largest cosine {float(sv.max()):.9f}, '
                      f'smallest {float(sv.min()):.3e}, closest two cosines of one pair {float((sv[:, :-1] - sv[:, 1:]).abs().min()):.3e}')
                derr = (got - dref.detach()).abs()
                k = int(derr.argmax())
                print(f'  worst distance entry {k}: got {float(got[k]):.7f} ref {float(dref[k]):.7f}')
            sys.exit(1)
    print(f'{cases} cases ok; worst rel err: ' + ', '.join(f'{k[0]}/{k[1]} {v:.1e}' for k, v in sorted(worst.items())))


if __name__ == '__main__':
    main()
