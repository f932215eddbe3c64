"""A/B helper: kernel-level timing (HIP events around the calls, graph-free) of SPD pdist fwd / fwd+bwd
for a few (d, n, dtype, spread) cases; run once per library via MM_MANIFOLDS_LIB."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'matrix-manifolds_amd'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import bench_configs as bc  # noqa: E402
from graphembed import manifolds as M  # noqa: E402

cases = [(3, 5000, torch.float32, 0.1), (3, 5000, torch.float32, 0.35), (3, 5000, torch.float64, 0.35),
         (4, 5000, torch.float32, 0.1), (4, 5000, torch.float32, 0.35), (4, 5000, torch.float64, 0.35),
         (4, 16384, torch.float32, 0.1), (2, 5000, torch.float32, 0.1), (5, 3000, torch.float32, 0.1)]
if len(sys.argv) > 1:
    cases = [c for c in cases if str(c[0]) in sys.argv[1]]
for d, n, dt, ir in cases:
    r = bc.pdist_case(M.SymmetricPositiveDefinite(d), n, dt, ir=ir)
    print(f'spd{d} n={n} {str(dt)[6:]} ir={ir}: fwd {r["fwd_us"]:.1f} us  fwd+bwd {r["fwd_bwd_us"]:.1f} us', flush=True)
