"""RiemannianAdam vs golden traces of the reference optimizer: control flow on
CPU stand-ins, and through the HIP manifolds on the GPU."""
import itertools

import numpy as np
import pytest
import torch

from conftest import load_golden

CASES = {'spd3': ('spd', 3), 'spd2': ('spd', 2), 'lorentz6': ('lorentz', 6), 'sphere6': ('sphere', 6),
         'euclidean10': ('euclidean', 10), 'grassmann52': ('grassmann', 5, 2), 'stiefel52': ('stiefel', 5, 2),
         'spd4': ('spd', 4), 'lorentz11': ('lorentz', 11)}


def run_trace(man, G, base, dname, to, tol):
    from graphembed.modules import ManifoldParameter
    from graphembed.optim import RiemannianAdam
    for exact, clip, nc in itertools.product([0, 1], [0, 1], [0, 1]):
        tag = f'{base}/exact{exact}_clip{clip}_nc{nc}'
        if f'{tag}/x1' not in G:
            continue
        p = ManifoldParameter(to(G[f'{base}/x0']), manifold=man)
        opt = RiemannianAdam([p], lr=0.05, betas=(0.9, 0.99), nc=bool(nc), max_grad_norm=2.0 if clip else None,
                             exact=bool(exact))
        for k in range(3):
            p.grad = to(G[f'{base}/g{k}'])
            opt.step()
            ref = G[f'{tag}/x{k + 1}']
            err = np.abs(p.data.double().cpu().numpy() - ref).max() / np.abs(ref).max()
            assert err <= tol, f'{tag}/x{k + 1}: {err:.2e}'
        assert float(opt.state[p]['step']) == 4.0  # advanced once per step (by the fused kernel on the GPU)
        ref = G[f'{tag}/exp_avg_sq']
        err = np.abs(opt.state[p]['exp_avg_sq'].double().cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-30)
        assert err <= tol * 10, f'{tag}/exp_avg_sq: {err:.2e}'
        ref = G[f'{tag}/exp_avg']
        err = np.abs(opt.state[p]['exp_avg'].double().cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-30)
        assert err <= tol * 10, f'{tag}/exp_avg: {err:.2e}'


@pytest.mark.parametrize('key', ['spd3', 'lorentz6', 'sphere6', 'euclidean10'])
def test_radam_control_flow_cpu(key):
    import cpu_double
    G = load_golden('radam')
    man = cpu_double.make(*CASES[key])
    run_trace(man, G, f'{key}/f64', 'f64', lambda a: torch.from_numpy(np.array(a)), 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize('key', list(CASES))
@pytest.mark.parametrize('dname', ['f32', 'f64'])
def test_radam_gpu(key, dname):
    from graphembed import manifolds as M
    G = load_golden('radam')
    kind = CASES[key]
    man = {'spd': M.SymmetricPositiveDefinite, 'lorentz': M.Lorentz, 'sphere': M.Sphere, 'euclidean': M.Euclidean,
           'grassmann': M.Grassmann, 'stiefel': M.Stiefel}[kind[0]](*kind[1:])
    run_trace(man, G, f'{key}/{dname}', dname, lambda a: torch.from_numpy(np.array(a)).cuda(),
              2e-4 if dname == 'f32' else 1e-7)


@pytest.mark.gpu
def test_radam_fused_kernel_is_used_and_flat_parameters():
    """Vector / SPD / flat parameters take the one-launch Adam kernels (mm_vec_radam_step, mm_spd_radam_step);
    a flat parameter's update equals torch's arithmetic of radam.py:62-98 with the Euclidean(1) fallback."""
    from graphembed import _backend as B
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldParameter
    from graphembed.optim import RiemannianAdam
    torch.manual_seed(3)
    xs = [ManifoldParameter(M.Lorentz(6).rand(20, out=torch.empty(0, device='cuda')), manifold=M.Lorentz(6)),
          ManifoldParameter(M.SymmetricPositiveDefinite(3).rand(20, out=torch.empty(0, device='cuda')),
                            manifold=M.SymmetricPositiveDefinite(3)),
          torch.nn.Parameter(torch.randn(5, 3, device='cuda', dtype=torch.float64))]
    w0 = xs[2].detach().clone()
    opt = RiemannianAdam(xs, lr=0.01, betas=(0.9, 0.99), max_grad_norm=0.5)
    lib, calls = B.lib(), []
    orig = lib.call
    lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    gs = []
    try:
        for _ in range(3):
            for p in xs:
                p.grad = torch.randn_like(p)
            gs.append(xs[2].grad.clone())
            opt.step()
    finally:
        del lib.call
    assert calls.count('mm_vec_radam_step') == 6 and calls.count('mm_spd_radam_step') == 3
    assert not any(c.endswith('_map') or c.endswith('_norm') for c in calls), calls
    w, m, v = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0)
    for t, g in enumerate(gs, start=1):
        nrm = (g * g).sum(-1, keepdim=True).clamp(min=1e-8).sqrt()
        gc = g * torch.clamp(0.5 / nrm, max=1.0)
        m = 0.9 * m + 0.1 * gc
        v = 0.99 * v + 0.01 * nrm**2
        alpha = 0.01 * (1 - 0.99**t)**0.5 / (1 - 0.9**t)
        w = w - alpha * m / (v.sqrt() + 1e-8)
    assert (xs[2].detach() - w).abs().max().item() <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize('dname', ['f32', 'f64'])
def test_radam_multi_parameter_launch_equals_single_steps(dname):
    """All vector-space parameters of a group in one launch (mm_vec_radam_step_multi): bit-identical points,
    moments and step counters to one fused launch per parameter, over 3 steps (incl. AdamNc)."""
    from graphembed import _backend as B
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldParameter
    from graphembed.optim import RiemannianAdam
    dt = {'f32': torch.float32, 'f64': torch.float64}[dname]
    for nc in (False, True):
        torch.manual_seed(5)
        mans = [M.Euclidean(5), M.Lorentz(6), M.Sphere(4)]
        pts = [man.rand(33 + 7 * k, out=torch.empty(0, dtype=dt, device='cuda')) for k, man in enumerate(mans)]
        scal = [torch.randn((), dtype=dt, device='cuda') for _ in range(2)]

        def build():
            ps = [ManifoldParameter(x.clone(), manifold=man) for man, x in zip(mans, pts)]
            return ps + [torch.nn.Parameter(s.clone()) for s in scal]
        pa, pb = build(), build()
        oa = RiemannianAdam(pa, lr=0.05, betas=(0.9, 0.99), nc=nc, max_grad_norm=1.0, exact=True)
        obs = [RiemannianAdam([p], lr=0.05, betas=(0.9, 0.99), nc=nc, max_grad_norm=1.0, exact=True) for p in pb]
        lib, calls = B.lib(), []
        orig = lib.call
        lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            for _ in range(3):
                gs = [torch.randn_like(p) for p in pa]
                for p, q, g in zip(pa, pb, gs):
                    p.grad, q.grad = g.clone(), g.clone()
                oa.step()
                for o in obs:
                    o.step()
        finally:
            del lib.call
        assert calls.count('mm_vec_radam_step_multi') == 3 and calls.count('mm_vec_radam_step') == 3 * len(pb)
        for p, q, o in zip(pa, pb, obs):
            assert torch.equal(p.detach(), q.detach())
            for key in ('exp_avg', 'exp_avg_sq', 'step'):
                assert torch.equal(oa.state[p][key], o.state[q][key]), key
            assert float(oa.state[p]['step']) == 4.0
