#!/usr/bin/env python3
"""What ONE rank's kernels cost when the headline problem is sharded N ways — measured on one GPU (no collective): the
graph-replayed prep + forward + backward + finalize of rank r's pair rows, for N = 1, 2, 4, 8 and every r.  The critical path of
an N-GPU step is the slowest rank's figure plus the all-reduce.  Also the config-5 size (n = 16384, SPD(4), fused loss step).
    python3 tools/shard_kernel_times.py > profiles/r02_shard_kernel_times.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402


def graph_us(wl, steps=40):
    fence = bench.Fence(1)
    graph, _ = bench.graph_of(wl.kernels, fence)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:      # warm clocks
        for _ in range(8):
            graph.replay()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e3


def main():
    dev = torch.device('cuda', 0)
    out = {'headline SPD(3) n=5000 f32 (us per step, kernels of one rank, hipGraph replay)': {},
           'config 5 SPD(4) n=16384 f32 fused QuotientLoss step (us)': {}}
    for world in (1, 2, 4, 8):
        ranks = []
        for r in range(world):
            wl = bench.PdistWorkload(3, 5000, torch.float32, 0.1, world, r, dev)
            ranks.append(round(graph_us(wl), 1))
            del wl
        out['headline SPD(3) n=5000 f32 (us per step, kernels of one rank, hipGraph replay)'][f'N={world}'] = ranks
    for world in (1, 2, 4, 8):
        ranks = []
        for r in sorted({0, world - 1}):
            wl = bench.FusedLossWorkload(4, 16384, torch.float32, world, r, dev)
            ranks.append(round(graph_us(wl, steps=10), 1))
            del wl
            torch.cuda.empty_cache()
        out['config 5 SPD(4) n=16384 f32 fused QuotientLoss step (us)'][f'N={world} (first, last rank)'] = ranks
    key = 'weak scaling SPD(3) f32, n = 5000 sqrt(N), 12.5 M pairs per rank (us, first and last rank)'
    out[key] = {}
    for world in (2, 4, 8):
        nw = int(round(5000 * world ** 0.5))
        ranks = []
        for r in sorted({0, world - 1}):
            wl = bench.PdistWorkload(3, nw, torch.float32, 0.1, world, r, dev, local_g=True)
            ranks.append(round(graph_us(wl), 1))
            del wl
            torch.cuda.empty_cache()
        out[key][f'N={world} (n={nw})'] = ranks
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
