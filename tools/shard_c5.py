#!/usr/bin/env python3
"""Config-5 shards (SPD(4), n = 16384, fused QuotientLoss step) of rank r of N on ONE GPU, graph-replayed: kernels of the step in us.
    python3 tools/shard_c5.py "2:0 2:1 4:0 4:3 8:0 8:7"      (MM_MANIFOLDS_LIB selects a library variant)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from shard_kernel_times import graph_us  # noqa: E402

dev = torch.device('cuda', 0)
out = []
for spec in (sys.argv[1] if len(sys.argv) > 1 else '1:0 2:0 2:1 4:0 4:3 8:0 8:7').split():
    world, r = (int(v) for v in spec.split(':'))
    wl = bench.FusedLossWorkload(4, 16384, torch.float32, world, r, dev)
    t = [round(graph_us(wl, steps=30), 1) for _ in range(3)]
    out.append(f'{spec} rows {tuple(wl.rows)}: {t}')
    del wl
    torch.cuda.empty_cache()
print(os.environ.get('MM_MANIFOLDS_LIB', 'main').split('libmm_')[-1], '|', '; '.join(out))
