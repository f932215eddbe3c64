"""Golden vectors of SPD `pdist` under NON-DEFAULT eigenvalue clamps from the REAL reference (development container only).
    PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0 python tests/golden/gen_golden_clamps.py
SymmetricPositiveDefinite(n, wmin=.., wmax=..).pdist with gradients (graphembed/manifolds/spd.py:29-30, 163-181: the
eigenvalues of L_i^-1 X_j L_i^-T are value-clamped to [wmin, wmax] before the logarithm), n = 2, 3, 4, 6, fp32 + fp64, points
spread widely enough that the clamps bind on a good part of the pairs; output tests/golden/clamps.npz."""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from graphembed.manifolds import SymmetricPositiveDefinite  # noqa: E402
from gen_golden import np_, DT  # noqa: E402

WINDOWS = [(0.6, 1.5), (0.25, 1.35), (1e-3, 2.0)]


def main():
    out = {}
    for d in (2, 3, 4, 6):
        for dname in DT:
            torch.set_default_dtype(DT[dname])
            for wi, (wmin, wmax) in enumerate(WINDOWS):
                torch.manual_seed(zlib.crc32(repr((d, dname, wi, 'clamps')).encode()) % (2**31))
                man = SymmetricPositiveDefinite(d, wmin=wmin, wmax=wmax)
                n = 40
                # points exp(U), ||U||_F between 0.1 and 0.5: pair spectra from ~1 to ~e^(+-1)
                u = torch.randn(n, d * (d + 1) // 2)
                u = u / u.norm(dim=-1, keepdim=True) * (0.1 + 0.4 * torch.rand(n, 1))
                plain = SymmetricPositiveDefinite(d)
                x = plain.exp(torch.eye(d).expand(n, d, d).contiguous(), plain.from_vec(u))
                x = x.detach().clone().requires_grad_()
                g = torch.randn(n * (n - 1) // 2)
                tag = f'spd{d}/{dname}/w{wi}'
                out[f'{tag}/x'] = np_(x)
                out[f'{tag}/g'] = np_(g)
                out[f'{tag}/window'] = np.array([wmin, wmax])
                d2 = man.pdist(x, squared=True)
                gr, = torch.autograd.grad((d2 * g).sum(), x)
                out[f'{tag}/d2'] = np_(d2)
                out[f'{tag}/grad_d2'] = np_(gr)
                free = plain.pdist(x.detach(), squared=True)
                out[f'{tag}/bound_fraction'] = np.array(float(((free - d2.detach()).abs() > 1e-4 * free.abs()).double().mean()))
    np.savez_compressed(os.path.join(HERE, 'clamps.npz'), **out)
    print(len(out), 'arrays;', {k: float(v) for k, v in out.items() if k.endswith('bound_fraction')})
    torch.set_default_dtype(torch.float32)


if __name__ == '__main__':
    main()
