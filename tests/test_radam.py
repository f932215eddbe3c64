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
