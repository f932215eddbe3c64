import sys, time, torch
sys.path.insert(0, '/root/repo/matrix-manifolds_amd')
from graphembed import manifolds as M
for (N, p, n) in ((6, 3, 2000), (9, 4, 2000)):
    for dt in (torch.float32, torch.float64):
        man = M.Grassmann(N, p)
        x = torch.linalg.qr(torch.randn(n, N, p, dtype=torch.float64))[0].to(dt).cuda().requires_grad_()
        g = torch.randn(n * (n - 1) // 2, dtype=dt, device='cuda')
        for _ in range(3):
            d = man.pdist(x, squared=True); gr, = torch.autograd.grad(d, x, g)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            d = man.pdist(x, squared=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(10):
            d = man.pdist(x, squared=True); gr, = torch.autograd.grad(d, x, g)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f'Gr({N},{p}) n={n} {dt}: fwd {(t1 - t0) / 10 * 1e6:.0f} us, fwd+bwd {(t2 - t1) / 10 * 1e6:.0f} us')
