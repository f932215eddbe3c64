#!/usr/bin/env python3
"""Randomised cross-check of the mixed-manifold pair kernel (mm_product_pairs_loss[_subset]) against the
differentiable per-factor path (compute_dists -> objective -> autograd): random sizes (incl. n = 2, tile
edges), factor mixes and dimensions, row shards, node subsets, both losses, fp32/fp64.
Usage: python tools/fuzz_product.py [cases] [seed] [--single] [--big]   (--single: one factor, the specialised fused kernels)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'matrix-manifolds_amd'))
import torch  # noqa: E402
from graphembed import _backend as B  # noqa: E402
from graphembed import manifolds as M  # noqa: E402
from graphembed.modules import ManifoldEmbedding, _pair_kernel_factors, _single_subset_factor  # noqa: E402
from graphembed.objectives import QuotientLoss, StressLoss  # noqa: E402


def squareform(v, n):
    d = torch.zeros(n, n, dtype=v.dtype, device=v.device)
    iu = torch.triu_indices(n, n, 1, device=v.device)
    d[iu[0], iu[1]] = v
    return d + d.t()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    cases = int(args[0]) if args else 200
    rng = random.Random(int(args[1]) if len(args) > 1 else 0)
    worst = {torch.float32: 0.0, torch.float64: 0.0}
    for c in range(cases):
        dt = rng.choice([torch.float32, torch.float64])
        n = rng.choice([2, 3, 5, 17, 63, 64, 65, 127, 129, 200, 257, rng.randint(2, 400)])
        if '--big' in sys.argv:   # more rows per wavefront (the kernel sizes them by n), many workgroups
            n = rng.choice([600, 777, 1025, 1500, 2100, 3001] + ([4200, 5003] if '--single' in sys.argv else []))
        single = '--single' in sys.argv
        nv = rng.randint(0, 3)
        mans = []
        if single:   # one factor: the specialised fused kernels (mm_spd_pdist_loss, mm_vec_pdist_loss / Gram loss)
            fam = rng.choice(['spd', 'e', 'l', 's'])
            if fam == 'spd':
                mans = [M.SymmetricPositiveDefinite(rng.choice([2, 3, 3, 4, 4, 5, 6]))]   # (6: rolled Jacobi sweeps)
            else:
                man = {'e': M.Euclidean, 'l': M.Lorentz, 's': M.Sphere}[fam](rng.choice([rng.randint(2, 32), rng.randint(17, 64)]))
                man.use_gram = fam != 'e' and rng.random() < 0.5
                mans = [man]
            nv = -1
        for _ in range(nv):
            kind = rng.choice(['e', 'l', 's'])
            m = rng.randint(2, 16)
            mans.append({'e': M.Euclidean, 'l': M.Lorentz, 's': M.Sphere}[kind](m))
        if not single:
            if rng.random() < 0.7 or not mans:
                mans.append(M.SymmetricPositiveDefinite(rng.choice([2, 3])))
            rng.shuffle(mans)
            if len(mans) < 2:
                mans.append(M.Euclidean(rng.randint(1, 16)))
            assert _pair_kernel_factors(mans) is not None
        torch.manual_seed(c)
        torch.set_default_dtype(dt)
        try:
            with torch.device('cuda'):
                emb = ManifoldEmbedding(n, mans)
                with torch.no_grad():
                    emb.perturb(rng.choice([0.05, 0.3, 0.8]))
                    for s in emb.scales:
                        s.fill_(rng.uniform(-1.0, 2.0))
        finally:
            torch.set_default_dtype(torch.float32)
        params = list(emb.xs) + list(emb.scales)
        md = emb.compute_dists(None).detach()
        P = md.numel()
        target = md * (0.4 + 1.2 * torch.rand(P, dtype=dt, device='cuda')) + 0.05
        fn, kw = (StressLoss(), {}) if rng.random() < 0.5 else (QuotientLoss(), dict(epoch=rng.randint(0, 5), alpha=rng.uniform(0.7, 1.4)))
        # node minibatch: through the mixed-manifold pair kernel, or — a single factor it does not take (SPD(4...), vectors wider
        # than 16) — through the factor's own pair kernel in its SUB form (mm_spd_pdist_loss_subset / mm_vec_pdist_loss_subset)
        subset = n >= 4 and rng.random() < 0.5 and (_pair_kernel_factors(mans) is not None or
                                                     (single and _single_subset_factor(mans[0]) is not None))
        if subset:
            bs = rng.randint(2, min(n, 2048))   # (larger batches are not handled inside the kernel: fused_objective -> None)
            idx = torch.randperm(n, device='cuda')[:bs]
            dense = squareform(target, n)
            iu = torch.triu_indices(bs, bs, 1, device='cuda')
            tsub = dense[idx][:, idx][iu[0], iu[1]]
            ref = fn(tsub, emb.compute_dists(idx), **kw)
            world = rng.randint(1, 3)
            parts = [emb.fused_objective(fn, None, idx, rows=B.shard_rows(bs, world, r), dense=dense, **kw)
                     for r in range(world)]
        else:
            ref = fn(target, emb.compute_dists(None), **kw)
            world = rng.randint(1, 3)
            parts = []
            for r in range(world):
                rows = B.shard_rows(n, world, r)
                lo, hi = B.pair_offset(n, rows[0]), B.pair_offset(n, rows[1])
                parts.append(emb.fused_objective(fn, target[lo:hi], None, rows=rows, **kw))
        rg = torch.autograd.grad(ref, params)
        tot = sum(p.item() for p in parts)
        gs = [sum(g) for g in zip(*[torch.autograd.grad(p, params) for p in parts])]
        tol = 3e-4 if dt == torch.float32 else 1e-9
        # the quotient loss has kinks: a pair that sits within rounding of one flips a +-1 — compare with slack
        kink = isinstance(fn, QuotientLoss)
        # (a loss that nearly vanishes — case 793 of seed 57007: ONE pair whose distance matches its target to 2e-3 — is a
        # difference of nearly equal numbers: measured against the size of its terms)
        le = abs(tot - ref.item()) / max(abs(ref.item()), 1e-2 * float((target.double() ** 2).sum()) if not subset else 0.0, 1e-30)
        # (a gradient that cancels to ~0 — a scale's — is measured against the largest gradient of the step)
        gmax = max(float(b.abs().max()) for b in rg)
        ge = max(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-3 * gmax, 1e-30) for a, b in zip(gs, rg))
        worst[dt] = max(worst[dt], le, 0.0 if kink else ge)
        ok = le <= tol and (ge <= (50 * tol if not kink else 0.2))
        if not ok or not all(bool(torch.isfinite(g).all()) for g in gs):
            for a, b in zip(gs, rg):
                print('  param', tuple(b.shape), 'max|ref|', float(b.abs().max()), 'max|got|', float(a.abs().max()),
                      'max|diff|', float((a - b).abs().max()))
            print('  target', target.tolist()[:6], 'md', md.tolist()[:6], 'kw', kw)
            print(f'FAIL case {c}: n={n} mans={[str(m) for m in mans]} dt={dt} subset={subset} world={world} '
                  f'loss={type(fn).__name__} le={le:.2e} ge={ge:.2e}')
            sys.exit(1)
    print(f'{cases} cases ok; worst rel err fp32 {worst[torch.float32]:.2e}, fp64 {worst[torch.float64]:.2e}')


if __name__ == '__main__':
    main()
