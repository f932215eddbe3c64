"""Host-side cost of one eager training step (fused loss + RSGD): cProfile over 300 steps.
Usage: python tools/host_profile.py [product]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'matrix-manifolds_amd'))
import torch  # noqa: E402

from graphembed import manifolds as M  # noqa: E402
from graphembed.modules import ManifoldEmbedding  # noqa: E402
from graphembed.objectives import StressLoss  # noqa: E402
from graphembed.optim import RiemannianSGD  # noqa: E402

from graphembed import unit_seed  # noqa: E402

PRODUCT = len(sys.argv) > 1 and sys.argv[1] == 'product'
n = 1025 if PRODUCT else 2000
torch.manual_seed(0)
with torch.device('cuda'):
    emb = ManifoldEmbedding(n, [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)] if PRODUCT
                            else [M.SymmetricPositiveDefinite(3)])
target = torch.rand(n * (n - 1) // 2, device='cuda') * 0.99 + 0.01
opt = RiemannianSGD(list(emb.xs), lr=1e-3, exact=True, max_grad_norm=20)
opt_s = RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)
fn = StressLoss()


def step():
    opt.zero_grad()
    opt_s.zero_grad()
    loss = emb.fused_objective(fn, target, None)
    loss.backward(unit_seed(loss))
    opt.step()
    opt_s.step()


for _ in range(20):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(300):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host time per step (no sync): %.1f us' % ((t1 - t0) / 300 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
