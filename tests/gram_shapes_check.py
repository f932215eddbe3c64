"""Run by tests/test_round3_gpu.py in a subprocess with MM_GRAM_BWD_NW = 3 / 4 (the variable is read once per process):
the symmetric-tile matrix-core backward (csrc/vec_gram.hip, vec_gram_bwd_sym_f32_kernel) with super-tiles of 4 x 3 and of
4 x 4 tiles against the fp64 checker oracle/exact.c — plain backward, row shards summed, and the fused loss."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)


def main():
    from graphembed import _backend as B
    from graphembed import manifolds as M
    from graphembed.objectives import StressLoss
    from oracle import exact
    from oracle import ref_port as rp
    worst = 0.0
    for kind, m, n in (('lorentz', 11, 33), ('lorentz', 11, 700), ('sphere', 6, 417), ('lorentz', 6, 1025), ('sphere', 3, 129),
                       ('lorentz', 11, 2100)):
        gen = torch.Generator().manual_seed(n + m)
        x64 = rp.make(kind, m).rand(n, ir=0.3, dtype=torch.float64, generator=gen)
        g64 = torch.randn(n * (n - 1) // 2, dtype=torch.float64, generator=gen)
        xin = x64.float()
        ref = exact.vec_pdist_grad(kind, xin.double().numpy(), g64.float().double().numpy())
        man = {'lorentz': M.Lorentz, 'sphere': M.Sphere}[kind](m)
        x = xin.cuda().requires_grad_()
        gr, = torch.autograd.grad(man.pdist(x, squared=True), x, g64.float().cuda())
        err = np.abs(gr.double().cpu().numpy() - ref).max() / np.abs(ref).max()
        worst = max(worst, err)
        assert err <= 2e-4, (kind, m, n, err)
        # row shards of three ranks add up to the same gradient
        total = torch.zeros_like(gr)
        for r in range(3):
            rb, re = B.shard_rows(n, 3, r)
            lo, hi = B.pair_offset(n, rb), B.pair_offset(n, re)
            part, = torch.autograd.grad(man.pdist(x, squared=True, rows=(rb, re)), x, g64.float().cuda()[lo:hi])
            total += part
        err = (total - gr).abs().max().item() / gr.abs().max().item()
        assert err <= 2e-5, (kind, m, n, 'shards', err)
        # fused loss on the same kernel
        target = (torch.rand(n * (n - 1) // 2, generator=gen) * 0.9 + 0.05)
        s_raw = torch.tensor(0.3, device='cuda', requires_grad=True)
        xl = xin.cuda().requires_grad_()
        loss = man.pdist_loss(xl, s_raw, target.cuda(), StressLoss().fused_spec(), rows=(0, n))
        gx, = torch.autograd.grad(loss, xl)
        d2 = exact.vec_pdist(kind, xin.double().numpy())
        sp = float(np.log1p(np.exp(0.3)))
        res = sp * d2 - target.double().numpy()
        refl = exact.vec_pdist_grad(kind, xin.double().numpy(), 2 * res * sp)
        err = np.abs(gx.double().cpu().numpy() - refl).max() / np.abs(refl).max()
        assert err <= 3e-4, (kind, m, n, 'fused', err)
        assert abs(loss.item() - float((res ** 2).sum())) <= 1e-4 * float((res ** 2).sum())
    print(f'gram shapes ok (MM_GRAM_BWD_NW={os.environ.get("MM_GRAM_BWD_NW", "auto")}): worst grad rel err {worst:.2e}')


if __name__ == '__main__':
    main()
