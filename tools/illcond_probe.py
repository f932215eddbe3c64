"""fp32 accuracy of SPD pdist / its gradient on ILL-CONDITIONED points (cond(X) = 1e2 ... 1e6), against an fp64 evaluation of
the ROUNDED inputs (oracle/exact.c): what the eigen-solver route costs, not what rounding the input costs.
    python tools/illcond_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'matrix-manifolds_amd'))
from graphembed.manifolds import SymmetricPositiveDefinite as SPD  # noqa: E402
from oracle import exact  # noqa: E402


def points(n, d, cond, gen):
    q = torch.linalg.qr(torch.randn(n, d, d, dtype=torch.float64, generator=gen))[0]
    lam = torch.exp((torch.rand(n, d, dtype=torch.float64, generator=gen) - 0.5) * np.log(cond))
    lam[:, 0], lam[:, -1] = cond ** -0.5, cond ** 0.5
    return (q * lam.unsqueeze(1)) @ q.transpose(1, 2)


def main():
    gen = torch.Generator().manual_seed(0)
    n = 160
    for d in (2, 3, 4, 5):
        for cond in (1e2, 1e4, 1e6):
            x32 = points(n, d, cond, gen).float()
            x32 = 0.5 * (x32 + x32.transpose(1, 2))
            xin = x32.double().numpy()
            ref = exact.spd_pdist(xin)
            g = torch.randn(n * (n - 1) // 2, dtype=torch.float64, generator=gen)
            ref_g = exact.spd_pdist_grad(xin, g.numpy())
            x = x32.cuda().requires_grad_()
            d2 = SPD(d).pdist(x, squared=True)
            gr, = torch.autograd.grad(d2, x, g.float().cuda())
            e = np.abs(d2.detach().double().cpu().numpy() - ref) / np.abs(ref)
            ge = np.abs(gr.double().cpu().numpy() - ref_g).max() / np.abs(ref_g).max()
            print(f'SPD({d}) cond(X) = {cond:.0e}: d2 rel err max {e.max():.2e} median {np.median(e):.1e}; gradient {ge:.2e} of its largest entry')


if __name__ == '__main__':
    main()
