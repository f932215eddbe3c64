"""Host cost of the eager plugin path (Manifold.pdist + autograd backward, no graph): per-stage wall time without
synchronisation and a cProfile of 500 steps.   python tools/eager_profile.py [spd|euclid]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'matrix-manifolds_amd'))
import torch  # noqa: E402

from graphembed import manifolds as M  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'euclid'
if which == 'spd':
    man, n = M.SymmetricPositiveDefinite(3), 300
else:
    man, n = M.Euclidean(10), 40
torch.manual_seed(0)
x = man.rand(n, out=torch.empty(0, device='cuda')).requires_grad_()
g = torch.randn(n * (n - 1) // 2, device='cuda')


def fwd():
    return man.pdist(x, squared=True)


def both():
    d2 = man.pdist(x, squared=True)
    return torch.autograd.grad(d2, x, g)


def timed(f, k=2000):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / k * 1e6, (t2 - t0) / k * 1e6


print(which, 'n', n)
print('forward only      : host %.1f us / step, with final sync %.1f us' % timed(fwd))
print('forward + backward: host %.1f us / step, with final sync %.1f us' % timed(both))
from graphembed import _backend as B  # noqa: E402
print('host of the two calls:', 'C++ autograd nodes (lib/_mm_autograd.so)' if B.autograd_ext() is not None
      else 'torch.autograd.Function classes (MM_PY_AUTOGRAD=1 or the module is not built)')
with torch.autograd.set_multithreading_enabled(False):   # the engine runs the backward on the calling thread: no hand-off
    print('forward + backward, autograd multithreading off: host %.1f us / step, with final sync %.1f us' % timed(both))
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    both()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
