"""Golden vectors of the Stein-divergence path from the REAL reference (development container only).
    PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0 python tests/golden/gen_golden_stein.py
SymmetricPositiveDefinite(n, use_stein_div=True).pdist / .dist with gradients
(graphembed/manifolds/spd.py:183-194, 246-295), n = 2, 3, 4, fp32 + fp64, reference init and
well-conditioned random points; output tests/golden/stein.npz."""
import os
import zlib
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from graphembed.manifolds import SymmetricPositiveDefinite  # noqa: E402
from gen_golden import np_, DT  # noqa: E402


def main():
    out = {}
    for d in (2, 3, 4):
        for dname in DT:
            torch.set_default_dtype(DT[dname])
            torch.manual_seed(zlib.crc32(repr((d, dname, 'stein')).encode()) % (2**31))
            man = SymmetricPositiveDefinite(d, use_stein_div=True)
            for init in ('rand', 'wide'):
                for n in (33, 70):
                    if init == 'rand':
                        x = man.rand(n)
                    else:
                        a = torch.rand(n, d, d)
                        x = a @ a.transpose(1, 2) + torch.eye(d)
                    x = x.detach().clone().requires_grad_()
                    g = torch.randn(n * (n - 1) // 2)
                    tag = f'spd{d}/{dname}/{init}/n{n}'
                    out[f'{tag}/x'] = np_(x)
                    out[f'{tag}/g'] = np_(g)
                    for squared in (True, False):
                        div = man.pdist(x, squared=squared)
                        gr, = torch.autograd.grad((div * g).sum(), x)
                        sfx = 'sq' if squared else 'rt'
                        out[f'{tag}/div_{sfx}'] = np_(div)
                        out[f'{tag}/grad_{sfx}'] = np_(gr)
                    # element-wise form on consecutive pairs
                    y = x.detach().flip(0).clone().requires_grad_()
                    xx = x.detach().clone().requires_grad_()
                    dd = man.dist(xx, y, squared=True)
                    gx, gy = torch.autograd.grad(dd.sum(), [xx, y])
                    out[f'{tag}/dist_sq'] = np_(dd)
                    out[f'{tag}/dist_gx'] = np_(gx)
                    out[f'{tag}/dist_gy'] = np_(gy)
    np.savez_compressed(os.path.join(HERE, 'stein.npz'), **out)
    print(len(out), 'arrays')
    torch.set_default_dtype(torch.float32)


if __name__ == '__main__':
    main()
