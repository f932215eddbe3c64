"""Golden RiemannianAdam traces from the REAL reference (development container only).
    PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0 python tests/golden/gen_golden_radam.py
Three consecutive steps with fixed Euclidean gradients, for every manifold on the path and the
(exact, clip, nc) settings; output tests/golden/radam.npz."""
import itertools
import os
import zlib
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from graphembed.modules import ManifoldParameter  # noqa: E402
from graphembed.optim import RiemannianAdam  # noqa: E402
from gen_golden import MANIFOLDS, make_points, np_, DT  # noqa: E402


def main():
    out = {}
    n = 17
    for key in ['spd2', 'spd3', 'spd4', 'lorentz11', 'lorentz6', 'sphere6', 'euclidean10', 'grassmann52', 'stiefel52']:
        for dname in DT:
            torch.set_default_dtype(DT[dname])
            torch.manual_seed(zlib.crc32(repr((key, dname, 'radam')).encode()) % (2**31))
            man = MANIFOLDS[key][0]()
            x0 = make_points(key, man, n, 'wide').detach().clone()
            gs = [torch.randn_like(x0) * 3.0 for _ in range(3)]
            if key.startswith('spd'):
                gs = [g + g.transpose(-2, -1) for g in gs]
            base = f'{key}/{dname}'
            out[f'{base}/x0'] = np_(x0)
            for k, g in enumerate(gs):
                out[f'{base}/g{k}'] = np_(g)
            for exact, clip, nc in itertools.product([False, True], [None, 2.0], [False, True]):
                if exact and key.startswith('stiefel'):
                    continue
                p = ManifoldParameter(x0.clone(), manifold=man)
                opt = RiemannianAdam([p], lr=0.05, betas=(0.9, 0.99), nc=nc, max_grad_norm=clip, exact=exact)
                tag = f'{base}/exact{int(exact)}_clip{0 if clip is None else 1}_nc{int(nc)}'
                for k, g in enumerate(gs):
                    p.grad = g.clone()
                    opt.step()
                    out[f'{tag}/x{k + 1}'] = np_(p.data)
                out[f'{tag}/exp_avg'] = np_(opt.state[p]['exp_avg'])
                out[f'{tag}/exp_avg_sq'] = np_(opt.state[p]['exp_avg_sq'])
    np.savez_compressed(os.path.join(HERE, 'radam.npz'), **out)
    print(len(out), 'arrays')
    torch.set_default_dtype(torch.float32)


if __name__ == '__main__':
    main()
