"""Pins oracle.ref_port (the reference-faithful CPU port) against the golden
vectors recorded from the real reference (tests/golden/gen_golden.py).  CPU only."""
import itertools

import numpy as np
import pytest
import torch

from conftest import load_golden, sym
from oracle import ref_port as rp

DT = {'f32': torch.float32, 'f64': torch.float64}
# (rtol on max|ref|) per dtype: the port repeats the reference's op sequence,
# so fp64 agrees to rounding; fp32 to a few ulps amplified by acos'/1/gap terms.
TOL = {'f64': 1e-9, 'f32': 2e-4}

MANS = {
    'spd2': ('spd', 2), 'spd3': ('spd', 3), 'spd4': ('spd', 4), 'spd5': ('spd', 5),
    'spd6': ('spd', 6), 'spd7': ('spd', 7), 'spd8': ('spd', 8), 'spd9': ('spd', 9),
    'lorentz11': ('lorentz', 11), 'lorentz6': ('lorentz', 6), 'lorentz3': ('lorentz', 3),
    'lorentz48': ('lorentz', 48), 'sphere64': ('sphere', 64), 'euclidean40': ('euclidean', 40),
    'sphere6': ('sphere', 6), 'euclidean10': ('euclidean', 10),
    'grassmann52': ('grassmann', 5, 2), 'grassmann63': ('grassmann', 6, 3),
    'stiefel52': ('stiefel', 5, 2),
}


def close(a, b, tol, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    # the reference itself yields NaN/inf in places (e.g. Grassmann fp32 at its
    # own init: acos'(1)); those entries are "don't care", the rest must agree
    ok = np.isfinite(b)
    if ok.mean() < 0.5:  # the reference has no answer here (all-NaN gradient)
        return
    a, b = a[ok], b[ok]
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= tol, f'{what}: rel-to-max err {err:.3e} > {tol:.1e}'


def T(a):
    return torch.from_numpy(np.array(a))


@pytest.mark.parametrize('key', [k for k in MANS if not k.startswith('stiefel')])
@pytest.mark.parametrize('dname,init', list(itertools.product(DT, ['rand', 'wide'])))
def test_pdist_and_grad(key, dname, init):
    G = load_golden(key)
    man = rp.make(*MANS[key])
    for n in (33, 96):
        tag = f'{dname}/{init}/n{n}'
        if f'{tag}/x' not in G:
            continue
        tol = TOL[dname]
        if dname == 'f32' and init == 'rand' and key.startswith('grassmann'):
            tol *= 50  # acos(sigma ~ 1) in fp32: LAPACK-rounding-level changes in sigma show
        x, g = T(G[f'{tag}/x']), T(G[f'{tag}/g'])
        xr = x.clone().requires_grad_()
        d2 = man.pdist(xr, squared=True)
        close(d2.detach(), G[f'{tag}/d2'], tol, 'd2')
        gr = torch.autograd.grad((d2 * g).sum(), xr)[0]
        # spectra of random-init SPD points are nearly degenerate: the
        # reference's acos' amplifies rounding there
        gtol = tol * (50 if (dname == 'f32' and key.startswith('spd')) else 1)
        ref = G[f'{tag}/grad_d2']
        if key.startswith('spd'):
            close(sym(gr.numpy()), sym(ref), gtol, 'sym grad_d2')
        else:
            close(gr, ref, gtol, 'grad_d2')
        xr = x.clone().requires_grad_()
        d1 = man.pdist(xr, squared=False)
        close(d1.detach(), G[f'{tag}/d1'], tol, 'd1')
        gr = torch.autograd.grad((d1 * g).sum(), xr)[0]
        ref = G[f'{tag}/grad_d1']
        if key.startswith('spd'):
            close(sym(gr.numpy()), sym(ref), gtol, 'sym grad_d1')
        else:
            close(gr, ref, gtol, 'grad_d1')
        close(man.dist(x, x.flip(0), squared=True), G[f'{tag}/dist_xy'], tol, 'dist_xy')


@pytest.mark.parametrize('key', list(MANS))
@pytest.mark.parametrize('dname,init', list(itertools.product(DT, ['rand', 'wide'])))
def test_optimizer_side_maps(key, dname, init):
    G = load_golden(key)
    man = rp.make(*MANS[key])
    tag = f'{dname}/{init}/n33'
    tol = TOL[dname] * (20 if dname == 'f32' else 1)
    x = T(G[f'{tag}/x'])
    eg = T(G[f'{tag}/grad_d2'] if f'{tag}/grad_d2' in G else G[f'{tag}/egrad_in'])
    rg = man.egrad2rgrad(x, eg)
    close(rg, G[f'{tag}/rgrad'], tol, 'egrad2rgrad')
    close(man.norm(x, rg, keepdim=True), G[f'{tag}/rgrad_norm'], tol, 'norm')
    u = T(G[f'{tag}/u'])
    pu = man.proju(x, u)
    close(pu, G[f'{tag}/proju'], tol, 'proju')
    close(man.retr(x, pu), G[f'{tag}/retr'], tol, 'retr')
    if not key.startswith('stiefel'):
        close(man.exp(x, pu), G[f'{tag}/exp'], tol, 'exp')
        close(man.log(x, x.flip(0)), G[f'{tag}/log'], tol * 10, 'log')
        close(man.projx(T(G[f'{tag}/projx_in'])), G[f'{tag}/projx'], tol, 'projx')
    else:
        close(man.orthonormalize(T(G[f'{tag}/projx_in'])), G[f'{tag}/projx'], tol, 'orthonormalize')
        close(man.retr_qr(x, pu), G[f'{tag}/retr_qr'], tol, 'retr_qr')
    close(man.transp(x, man.retr(x, pu), pu), G[f'{tag}/transp'], tol, 'transp')


@pytest.mark.parametrize('key', list(MANS))
@pytest.mark.parametrize('dname', list(DT))
def test_rsgd_steps(key, dname):
    G = load_golden(key)
    man = rp.make(*MANS[key])
    base = f'{dname}/rsgd'
    tol = TOL[dname] * (20 if dname == 'f32' else 1)
    x0, g1, g2 = T(G[f'{base}/x0']), T(G[f'{base}/g1']), T(G[f'{base}/g2'])
    for exact, clip, mom in itertools.product([0, 1], [0, 1], [0, 1]):
        tag = f'{base}/exact{exact}_clip{clip}_mom{mom}'
        if f'{tag}/x1' not in G:
            continue
        kw = dict(lr=0.05, momentum=0.9 if mom else 0.0, dampening=0.1 if mom else 0.0,
                  max_grad_norm=2.0 if clip else None, exact=bool(exact))
        x1, buf = rp.rsgd_step(man, x0, g1, **kw)
        close(x1, G[f'{tag}/x1'], tol, tag + '/x1')
        x2, buf = rp.rsgd_step(man, x1, g2, momentum_buffer=buf, **kw)
        close(x2, G[f'{tag}/x2'], tol, tag + '/x2')
        if mom:
            close(buf, G[f'{tag}/buf2'], tol, tag + '/buf2')


@pytest.mark.parametrize('dname', list(DT))
def test_product_and_losses(dname):
    G = load_golden('callers')
    mans = [rp.Lorentz(6), rp.Sphere(6), rp.SPD(2)]
    tol = TOL[dname] * (20 if dname == 'f32' else 1)
    idx = T(G[f'{dname}/product/idx'])
    for tag, ii in [('full', None), ('batch', idx)]:
        base = f'{dname}/product/{tag}'
        xs = [T(G[f'{base}/x_{k}']).requires_grad_() for k in range(3)]
        sc = [torch.tensor(float(s), dtype=DT[dname], requires_grad=True)
              for s in G[f'{base}/scales']]
        md = rp.compute_dists(mans, xs, sc, ii)
        close(md.detach(), G[f'{base}/d2'], tol, 'product d2')
        grads = torch.autograd.grad((md * T(G[f'{base}/g'])).sum(), xs + sc)
        for k in range(3):
            a, b = grads[k].numpy(), G[f'{base}/grad_x_{k}']
            if k == 2:
                a, b = sym(a), sym(b)
            close(a, b, tol, f'product grad_x_{k}')
            close(grads[3 + k], G[f'{base}/grad_s_{k}'], tol, f'product grad_s_{k}')
    gd, md = T(G[f'{dname}/loss/gd']), T(G[f'{dname}/loss/md'])
    close(rp.stress_loss(gd, md), G[f'{dname}/loss/stress'], tol, 'stress')
    close(rp.quotient_loss(gd, md, epoch=3, alpha=1.7), G[f'{dname}/loss/quotient'], tol, 'quotient')


@pytest.mark.parametrize('case', ['euclidean10', 'lorentz11', 'spd3', 'product'])
@pytest.mark.parametrize('loss_name', ['stress', 'quotient'])
def test_tree40_training_trace(case, loss_name):
    """20 full-batch epochs on tree40 (config[0] plumbing), fp64, the reference's
    production RSGD settings (experiments/run_grid.py:24-36)."""
    G = load_golden('callers')
    mk = {'euclidean10': lambda: [rp.Euclidean(10)], 'lorentz11': lambda: [rp.Lorentz(11)],
          'spd3': lambda: [rp.SPD(3)],
          'product': lambda: [rp.Lorentz(6), rp.Sphere(6), rp.SPD(2)]}[case]
    mans = mk()
    base = f'tree40/{case}/{loss_name}'
    target = T(G['tree40/target'])
    xs = [T(G[f'{base}/x0_{k}']) for k in range(len(mans))]
    sc = [torch.tensor(0.5, dtype=torch.float64) for _ in mans]
    eu = rp.Euclidean(1)
    losses = []
    for epoch in range(20):
        xr = [x.clone().requires_grad_() for x in xs]
        sr = [s.clone().requires_grad_() for s in sc]
        md = rp.compute_dists(mans, xr, sr)
        loss = (rp.stress_loss(target, md) if loss_name == 'stress' else
                rp.quotient_loss(target, md, epoch=epoch, alpha=1.0))
        grads = torch.autograd.grad(loss, xr + sr)
        losses.append(loss.item())
        xs = [rp.rsgd_step(m, x, g, lr=0.01, exact=True, max_grad_norm=20)[0]
              for m, x, g in zip(mans, xs, grads[:len(mans)])]
        sc = [rp.rsgd_step(eu, s, g, lr=1e-4, max_grad_norm=500)[0]
              for s, g in zip(sc, grads[len(mans):])]
    close(np.array(losses), G[f'{base}/losses'], 1e-6, 'loss trace')
    for k in range(len(mans)):
        close(xs[k], G[f'{base}/x20_{k}'], 1e-6, f'x20_{k}')


@pytest.mark.parametrize('d', [2, 3, 4])
@pytest.mark.parametrize('dname', ['f32', 'f64'])
def test_stein_divergence(d, dname):
    """The port's Stein path (spd.py:183-194, 246-295) against vectors recorded from the reference's
    PairwiseSteinDivergence / stein_div, values and autograd gradients."""
    G = load_golden('stein')
    man = rp.SPD(d)
    tol = 2e-4 if dname == 'f32' else 1e-9
    for init in ('rand', 'wide'):
        for n in (33, 70):
            tag = f'spd{d}/{dname}/{init}/n{n}'
            x = T(G[f'{tag}/x']).requires_grad_()
            g = T(G[f'{tag}/g'])
            for squared, sfx in ((True, 'sq'), (False, 'rt')):
                div = man.stein_pdiv(x, squared=squared)
                close(div.detach(), G[f'{tag}/div_{sfx}'], tol, f'div {tag} {sfx}')
                gr, = torch.autograd.grad((div * g).sum(), x)
                close(sym(gr.numpy()), sym(G[f'{tag}/grad_{sfx}']), tol * 10, f'grad {tag} {sfx}')
            xx = T(G[f'{tag}/x']).requires_grad_()
            y = T(G[f'{tag}/x']).flip(0).clone().requires_grad_()
            dd = man.stein_div(xx, y, squared=True)
            close(dd.detach(), G[f'{tag}/dist_sq'], tol, f'dist {tag}')
            gx, gy = torch.autograd.grad(dd.sum(), [xx, y])
            close(sym(gx.numpy()), sym(G[f'{tag}/dist_gx']), tol * 10, f'dist gx {tag}')
            close(sym(gy.numpy()), sym(G[f'{tag}/dist_gy']), tol * 10, f'dist gy {tag}')


@pytest.mark.parametrize('d', [2, 3, 4, 6])
@pytest.mark.parametrize('dname', list(DT))
def test_spd_pdist_under_custom_eigenvalue_clamps(d, dname):
    """SymmetricPositiveDefinite(n, wmin=.., wmax=..) (spd.py:29-30, 163-169): eigenvalues of L_i^-1 X_j L_i^-T value-clamped to the
    window, d^2 value-clamped at wmin — golden from the real reference (tests/golden/gen_golden_clamps.py), three windows; pins
    the port and the exact fp64 checker (oracle/exact.c) for non-default clamps."""
    from oracle import exact
    G = load_golden('clamps')
    for wi in range(3):
        tag = f'spd{d}/{dname}/w{wi}'
        wmin, wmax = (float(v) for v in G[f'{tag}/window'])
        x, g = T(G[f'{tag}/x']), T(G[f'{tag}/g'])
        man = rp.SPD(d, wmin=wmin, wmax=wmax)
        xr = x.clone().requires_grad_()
        d2 = man.pdist(xr, squared=True)
        close(d2.detach(), G[f'{tag}/d2'], TOL[dname], f'{tag} d2')
        gr, = torch.autograd.grad((d2 * g).sum(), xr)
        close(sym(gr.detach().numpy()), sym(G[f'{tag}/grad_d2']), TOL[dname] * 5, f'{tag} grad')
        if dname == 'f64':
            xin, gin = x.double().numpy(), g.double().numpy()
            # (d = 2, 3: the reference's closed-form eigenvalues carry eps terms, linalg/fast.py:53-91 — SURVEY App. C)
            close(exact.spd_pdist(xin, wmin=wmin, wmax=wmax), G[f'{tag}/d2'], 1e-9 if d >= 4 else 5e-6, f'{tag} exact d2')
            close(exact.spd_pdist_grad(xin, gin, wmin=wmin, wmax=wmax), sym(G[f'{tag}/grad_d2']), 1e-8 if d >= 4 else 5e-6,
                  f'{tag} exact grad')
