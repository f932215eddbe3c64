#!/usr/bin/env python3
"""Per-rank kernel time of BASELINE config 5 (SPD(4), n = 16384, fused QuotientLoss step) under a shard-cut policy, measured on
one GPU (every rank's rows in turn, no collective).  The cut policy (MM_SHARD_K) and the column count of the SPD(4) backward
(MM_SPD4_BWD_TWO_COLS) are read once per process, so the caller sets them per run:
    MM_SHARD_K=0 MM_SPD4_BWD_TWO_COLS=0 python3 tools/shard_balance.py 8
prints one JSON line {world, K, two_cols, rows, pairs, us}."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from shard_kernel_times import graph_us  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device('cuda', 0)
    rows, pairs = [], []
    times = [[] for _ in range(world)]
    # every rank `rounds` times, round-robin (a single pass per rank showed +-5 % between neighbours: the order of the
    # measurements, not the rows)
    for rnd in range(rounds):
        for r in range(world):
            wl = bench.FusedLossWorkload(4, n, torch.float32, world, r, dev)
            if rnd == 0:
                rows.append(list(wl.rows))
                pairs.append(wl.hi - wl.lo)
            times[r].append(round(graph_us(wl, steps=20), 1))
            del wl
            torch.cuda.empty_cache()
    med = [sorted(t)[len(t) // 2] for t in times]
    print(json.dumps({'world': world, 'n': n, 'K': os.environ.get('MM_SHARD_K', 'default'),
                      'two_cols': os.environ.get('MM_SPD4_BWD_TWO_COLS', 'default'), 'rows': rows, 'pairs': pairs, 'us_median': med,
                      'us_all': times}))


if __name__ == '__main__':
    main()
