#!/usr/bin/env python3
"""Randomised cross-check of the per-point kernels (egrad2rgrad, proju, projx, exp, retr, log, transp, norm and the
element-wise dist forward + backward) of every manifold of the path against the reference-faithful torch port
oracle/ref_port.py: random point counts (incl. 1 and tile edges), dimensions, tangent sizes from 1e-6 to O(1).
Not collected by pytest (run by hand on a GPU box): python tests/fuzz_maps.py [cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from graphembed import manifolds as M  # noqa: E402
from oracle import ref_port as rp  # noqa: E402


def rel(a, b):
    b = b.detach()
    return float((a.detach().double().cpu() - b).abs().max() / b.abs().max().clamp(min=1e-30))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = {}
    for c in range(cases):
        torch.manual_seed(c)
        dt = rng.choice([torch.float32, torch.float64])
        f32 = dt == torch.float32
        cnt = rng.choice([1, 2, 63, 64, 65, 127, 129, rng.randint(1, 400)])
        fam = rng.choice(['spd', 'lorentz', 'sphere', 'euclidean', 'grassmann', 'stiefel'])
        if fam == 'spd':
            d = rng.choice([2, 3, 4, 5])
            man, ref = M.SymmetricPositiveDefinite(d), rp.SPD(d)
            x = ref.rand(cnt, ir=rng.choice([0.1, 0.7]), dtype=torch.float64)
            y = ref.rand(cnt, ir=rng.choice([0.1, 0.7]), dtype=torch.float64)
            raw = torch.randn(cnt, d, d, dtype=torch.float64)
            raw = raw + raw.transpose(1, 2)
            what = f'spd{d}'
        elif fam in ('grassmann', 'stiefel'):
            N, p = rng.choice([(3, 1), (4, 2), (5, 2), (6, 3), (9, 4), (3, 3)])
            if fam == 'grassmann' and N == p:
                N += 1
            man = (M.Grassmann if fam == 'grassmann' else M.Stiefel)(N, p)
            ref = rp.make(fam, N, p)
            x = torch.linalg.qr(torch.randn(cnt, N, p, dtype=torch.float64))[0]
            y = torch.linalg.qr(torch.randn(cnt, N, p, dtype=torch.float64))[0]
            raw = torch.randn(cnt, N, p, dtype=torch.float64)
            what = f'{fam}({N},{p})'
        else:
            m = rng.randint(2, 32) if rng.random() < 0.8 else rng.randint(33, 64)   # (the kernels go to 64)
            man = {'lorentz': M.Lorentz, 'sphere': M.Sphere, 'euclidean': M.Euclidean}[fam](m)
            ref = rp.make(fam, m)
            x = ref.rand(cnt, ir=rng.choice([0.01, 0.5]), dtype=torch.float64)
            y = ref.rand(cnt, ir=rng.choice([0.01, 0.5]), dtype=torch.float64)
            if fam == 'sphere':
                x = torch.nn.functional.normalize(torch.randn(cnt, m, dtype=torch.float64), dim=-1)
                y = torch.nn.functional.normalize(torch.randn(cnt, m, dtype=torch.float64), dim=-1)
            raw = torch.randn(cnt, m, dtype=torch.float64)
            what = f'{fam}{m}'
        scale = rng.choice([1e-6, 1e-3, 0.1, 1.0])
        # inputs rounded to the kernel's dtype, the port computes in fp64 on the same numbers
        x, y, raw = x.to(dt).double(), y.to(dt).double(), raw.to(dt).double()
        u = (ref.proju(x, raw) * scale).to(dt).double()
        gx, gy, gu = x.to(dt).cuda(), y.to(dt).cuda(), u.to(dt).cuda()
        tol = 5e-4 if f32 else (2e-6 if fam == 'spd' else 1e-8)   # SPD: the reference's eps-fudged closed forms
        checks = []
        with torch.no_grad():
            checks.append(('proju', man.proju(gx, raw.to(dt).cuda()), ref.proju(x, raw)))
            if fam not in ('grassmann', 'stiefel'):
                checks.append(('egrad2rgrad', man.egrad2rgrad(gx, raw.to(dt).cuda()), ref.egrad2rgrad(x, raw)))
            checks.append(('norm', man.norm(gx, gu), ref.norm(x, u)))
            checks.append(('retr', man.retr(gx, gu), ref.retr(x, u)))
            if fam != 'stiefel':
                checks.append(('exp', man.exp(gx, gu), ref.exp(x, u)))
            if fam in ('spd', 'lorentz', 'sphere'):
                new = ref.exp(x, u)
                checks.append(('transp', man.transp(gx, new.to(dt).cuda(), gu), ref.transp(x, new.to(dt).double(), u)))
            if fam == 'spd':
                checks.append(('projx', man.projx(gx), ref.projx(x)))
            # log_x(y): skipped where it is ill-conditioned (near-antipodal sphere points, a principal angle near
            # pi/2, fp32 Lorentz points closer than the rounding of their inner product)
            log_ok = fam in ('spd', 'euclidean')
            if fam == 'sphere':
                cc = (x * y).sum(-1)
                log_ok = float((1 - cc * cc).min()) > (1e-2 if f32 else 1e-8)
            if fam == 'lorentz':
                log_ok = float(ref.dist(x, y).min()) > (5e-2 if f32 else 1e-6)
            if fam == 'grassmann':
                sv = torch.linalg.svdvals(x.transpose(1, 2) @ y)
                log_ok = float(sv.min()) > 0.1 and float((1 - sv.max()**2)) > (1e-2 if f32 else 1e-8)
            if log_ok:
                checks.append(('log', man.log(gx, gy), ref.log(x, y)))
        for name, got, want in checks:
            e = rel(got, want)
            # results of size ~scale: measured against the larger of result and input scale
            worst[(fam, name, 'f32' if f32 else 'f64')] = max(worst.get((fam, name, 'f32' if f32 else 'f64'), 0.0), e)
            if not (e <= tol and bool(torch.isfinite(got).all())):
                print(f'FAIL case {c}: {what} {name} cnt={cnt} {dt} scale={scale} err {e:.2e}')
                sys.exit(1)
        # element-wise distance forward + backward (not Stiefel: none in the reference)
        if fam != 'stiefel':
            xr, yr = x.clone().requires_grad_(), y.clone().requires_grad_()
            squared = rng.random() < 0.6
            dref = ref.dist(xr, yr, squared=squared)
            if float(dref.detach().min()) < (5e-2 if f32 else 1e-5):
                continue   # sqrt / acos' at ~0: ill-conditioned draw
            g = torch.randn(cnt, dtype=torch.float64)
            grx, gry = torch.autograd.grad(dref, (xr, yr), g)
            if fam == 'spd':
                grx, gry = 0.5 * (grx + grx.transpose(1, 2)), 0.5 * (gry + gry.transpose(1, 2))
            xg, yg = gx.clone().requires_grad_(), gy.clone().requires_grad_()
            dgot = man.dist(xg, yg, squared=squared)
            ggx, ggy = torch.autograd.grad(dgot, (xg, yg), g.to(dt).cuda())
            dtol = 2e-3 if f32 else (2e-6 if fam == 'spd' else 1e-7)
            gtol = ((0.3 if fam == 'grassmann' else 2e-2) if f32 else (2e-5 if fam == 'spd' else 1e-6))
            if fam == 'sphere' and f32:
                cc = (x * y).sum(-1)
                if float((1 - cc * cc).min()) < 1e-3:
                    continue
            ed, eg = rel(dgot, dref), max(rel(ggx, grx), rel(ggy, gry))
            worst[(fam, 'dist', 'f32' if f32 else 'f64')] = max(worst.get((fam, 'dist', 'f32' if f32 else 'f64'), 0.0), ed, eg)
            if not (ed <= dtol and eg <= gtol):
                print(f'FAIL case {c}: {what} dist cnt={cnt} {dt} squared={squared} d err {ed:.2e} grad err {eg:.2e}')
                sys.exit(1)
    print(f'{cases} cases ok')
    for k in sorted(worst):
        if worst[k] > (1e-5 if k[2] == 'f32' else 1e-10):
            print('  worst', k, f'{worst[k]:.1e}')


if __name__ == '__main__':
    main()
