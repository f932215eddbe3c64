#!/usr/bin/env python3
"""Randomised cross-check of the fused optimizer kernels (RSGD with and without momentum, Riemannian Adam;
single- and multi-parameter launches) against the reference-faithful torch port oracle/ref_port.py, in fp64:
random manifolds, dimensions, point counts, hyper-parameters, several consecutive steps.
Not collected by pytest (run by hand on a GPU box): python tests/fuzz_optim.py [cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from graphembed import manifolds as M  # noqa: E402
from graphembed.modules import ManifoldParameter  # noqa: E402
from graphembed.optim import RiemannianAdam, RiemannianSGD  # noqa: E402
from oracle import ref_port as rp  # noqa: E402

EPS = 1e-8


def ref_adam(man, x, g, state, *, lr, betas, nc, clip, exact):
    """optim/radam.py:62-98 on the port's manifolds."""
    if not state:
        state.update(step=1, m=torch.zeros_like(x), v=torch.zeros_like(x))
    beta1, beta2 = betas
    rg = man.egrad2rgrad(x, g)
    nrm = man.norm(x, rg, keepdim=True)
    if clip is not None:
        rg = rg * torch.clamp(clip / nrm, max=1.0)
    t = state['step']
    if nc:
        beta2 = 1 - 1 / t
    state['m'] = state['m'] * beta1 + (1 - beta1) * rg
    state['v'] = state['v'] * beta2 + (1 - beta2) * nrm.pow(2)
    alpha = lr * (1 - beta2**t)**0.5 / (1 - beta1**t)
    direction = -alpha * state['m'] / (state['v'].sqrt() + EPS)
    new_x = (man.exp if exact else man.retr)(x, direction)
    state['m'] = man.transp(x, new_x, state['m'])
    state['step'] = t + 1
    return new_x


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for c in range(cases):
        torch.manual_seed(c)
        k = rng.randint(1, 5)
        mans, refs, xs = [], [], []
        for _ in range(k):
            fam = rng.choice(['spd', 'euclidean', 'lorentz', 'sphere', 'flat'])
            cnt = rng.choice([1, 2, 63, 64, 65, 129, rng.randint(1, 300)])
            if fam == 'spd':
                d = rng.choice([2, 3, 4, 5])
                man, ref = M.SymmetricPositiveDefinite(d), rp.SPD(d)
                x = ref.rand(cnt, ir=rng.choice([0.1, 0.6]), dtype=torch.float64)
            elif fam == 'flat':
                man = ref = None
                x = torch.randn(rng.choice([(), (3, ), (4, 5)]), dtype=torch.float64)
            else:
                m = rng.randint(2, 32) if rng.random() < 0.8 else rng.randint(33, 64)   # (the kernels go to 64)
                man = {'euclidean': M.Euclidean, 'lorentz': M.Lorentz, 'sphere': M.Sphere}[fam](m)
                ref = rp.make(fam, m)
                x = ref.rand(cnt, ir=rng.choice([0.01, 0.5]), dtype=torch.float64)
            mans.append(man)
            refs.append(ref if ref is not None else rp.Euclidean(x.shape[-1] if x.ndim else 1))
            xs.append(x)
        adam = rng.random() < 0.5
        lr = rng.choice([1e-3, 1e-2, 0.1])
        clip = rng.choice([None, 0.05, 5.0])
        exact = rng.random() < 0.5
        momentum = 0.0 if adam else rng.choice([0.0, 0.0, 0.9])
        damp = rng.choice([0.0, 0.1]) if momentum else 0.0
        nc = adam and rng.random() < 0.3
        betas = (0.9, 0.99)
        params = [ManifoldParameter(x.cuda(), manifold=man) if man is not None else torch.nn.Parameter(x.cuda())
                  for x, man in zip(xs, mans)]
        opt = (RiemannianAdam(params, lr=lr, betas=betas, nc=nc, max_grad_norm=clip, exact=exact) if adam else
               RiemannianSGD(params, lr=lr, momentum=momentum, dampening=damp, max_grad_norm=clip, exact=exact))
        rx = [x.clone() for x in xs]
        rstate = [dict() for _ in xs]
        rbuf = [None] * k
        for step in range(3):
            gs = []
            for x, man in zip(rx, mans):
                g = torch.randn_like(x) * rng.choice([0.1, 3.0])
                if isinstance(man, M.SymmetricPositiveDefinite):
                    g = g + g.transpose(-2, -1)
                gs.append(g)
            for p, g in zip(params, gs):
                p.grad = g.cuda()
            opt.step()
            def _advance_reference():
                for i, (ref, g) in enumerate(zip(refs, gs)):
                    flat = mans[i] is None
                    xr = rx[i].reshape(-1, rx[i].shape[-1] if rx[i].ndim else 1) if flat else rx[i]
                    gr = g.reshape(xr.shape)
                    if adam:
                        new = ref_adam(ref, xr, gr, rstate[i], lr=lr, betas=betas, nc=nc, clip=clip, exact=exact)
                    else:
                        new, rbuf[i] = rp.rsgd_step(ref, xr, gr, lr=lr, momentum=momentum, dampening=damp,
                                                    max_grad_norm=clip, exact=exact, momentum_buffer=rbuf[i])
                    rx[i] = new.reshape(rx[i].shape)
            try:
                _advance_reference()
            except (RuntimeError, ValueError):   # e.g. the retraction left the SPD cone: torch's Cholesky raises
                break
            if not all(bool(torch.isfinite(r).all()) and r.abs().max() < 1e6 for r in rx):
                break   # the draw diverged in the reference arithmetic itself (huge unclipped steps)
            if any(isinstance(m_, M.SymmetricPositiveDefinite) and float(torch.linalg.cond(r).max()) > 1e5
                   for m_, r in zip(mans, rx)):
                break   # ... or walked into ill-conditioned SPD points (errors ~ cond * eps from there on)
            for p, r in zip(params, rx):
                err = ((p.detach().cpu() - r).abs().max() / r.abs().max().clamp(min=1e-30)).item()
                worst = max(worst, err)
                if not err <= 1e-6:   # the reference's eps-fudged closed forms (SPD(2,3)) bias it by ~1e-7
                    print(f'FAIL case {c} step {step}: {[str(m) for m in mans]} adam={adam} lr={lr} clip={clip} '
                          f'exact={exact} momentum={momentum} nc={nc} err={err:.2e} shape={tuple(r.shape)}')
                    sys.exit(1)
    print(f'{cases} cases ok; worst rel err {worst:.2e}')


if __name__ == '__main__':
    main()
