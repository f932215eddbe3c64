#!/usr/bin/env python3
"""Headline benchmark: pairwise manifold-dist/sec (fwd+bwd), SPD(3), 5k-node graph.

One step = one pass of the hot path over all pairs of the synthetic embedding:
    d2 = SPD(3).pdist(x, squared=True);  d2.backward(g)
(`g` a fixed random upstream gradient, so no loss is fused in — SURVEY.md §8d),
with the pair list sharded by rows across the ranks and, for N > 1, ONE RCCL
all-reduce(sum) of the embedding gradient inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 is launched by torch.distributed.run (one rank per GPU).  Rank 0 prints one
JSON line.  Inputs are resident in HBM before the timed region starts.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

N_NODES = 5000
DIM = 3
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synthetic_spd(n, d, seed, device):
    """SPD.rand (spd.py:201-208): X = expm(U), ||vec U|| = 0.1 — the reference's init."""
    gen = torch.Generator().manual_seed(seed)
    m = d * (d + 1) // 2
    u = torch.randn(n, m, generator=gen, dtype=torch.float64)
    u = u / u.norm(dim=-1, keepdim=True) * 0.1
    iu = torch.triu_indices(d, d)
    U = torch.zeros(n, d, d, dtype=torch.float64)
    U[:, iu[0], iu[1]] = u / 2 ** 0.5
    U[:, iu[1], iu[0]] = u / 2 ** 0.5
    k = torch.arange(d)
    U[:, k, k] *= 2 ** 0.5
    X = torch.linalg.matrix_exp(U)
    P = n * (n - 1) // 2
    g = torch.randn(P, generator=gen, dtype=torch.float32)
    return X.float().to(device), g.to(device)


def cpu_baseline(seed):
    """The oracle port (reference-faithful op sequence) on this host's cores, on a
    bounded sample: n=2500 nodes (3.1 M pairs) of the same workload, best of 3."""
    from oracle import ref_port
    ncpu = os.cpu_count() or 1
    n = 2500
    x, g = synthetic_spd(n, DIM, seed, 'cpu')
    man = ref_port.SPD(DIM)
    P = n * (n - 1) // 2
    best, best_threads = float('inf'), 1
    # torch's intra-op pool does not scale to hundreds of threads on these element-wise ops:
    # time a few pool sizes and report the fastest (the fairest baseline for this host)
    for threads in sorted({min(8, ncpu), min(32, ncpu), min(64, ncpu), ncpu}):
        torch.set_num_threads(threads)
        for it in range(3):
            xr = x.clone().requires_grad_()
            t0 = time.perf_counter()
            d2 = man.pdist(xr, squared=True)
            d2.backward(g)
            dt = time.perf_counter() - t0
            if it and dt < best:
                best, best_threads = dt, threads
    return {'value': P / best, 'unit': 'pairs/s', 'cores': best_threads, 'kind': 'port',
            'host_cpus': ncpu,
            'sample': f'SPD(3) fp32 reference-init, n={n} ({P} pairs), fwd+bwd, best of 2 per pool size '
                      f'(8/32/64/all threads), fastest pool reported; oracle/ref_port.py '
                      f'(torch CPU, reference op sequence)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--n', type=int, default=N_NODES)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='no HIP-event bracketing of the kernels')
    ap.add_argument('--graph', choices=['auto', 'on', 'off'], default='auto',
                    help='replay the step from a captured hipGraph (auto = on; falls back to eager launches '
                         'if capture is unavailable)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    ndev = torch.cuda.device_count()
    dev = torch.device('cuda', local_rank % max(ndev, 1))   # (one rank per GPU on the real node)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        # "nccl" IS RCCL on ROCm.  MM_BENCH_BACKEND=gloo exists only to exercise the N > 1 code
        # path on a single-GPU development box.
        backend = os.environ.get('MM_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from graphembed import _backend as B
    from graphembed.manifolds import SymmetricPositiveDefinite
    lib = B.lib()

    n = args.n
    x, g = synthetic_spd(n, DIM, 42, dev)           # replicated embedding, same on every rank
    rb, re = B.shard_rows(n, world, rank)
    lo, hi = B.pair_offset(n, rb), B.pair_offset(n, re)
    g_local = g[lo:hi].contiguous()
    del g
    man = SymmetricPositiveDefinite(DIM)
    x.requires_grad_()

    def step():
        x.grad = None
        d2 = man.pdist(x, squared=True, rows=(rb, re))
        d2.backward(g_local)
        if world > 1:
            dist.all_reduce(x.grad)               # the single collective of a step
        return d2

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The step is 4 kernels of 4-70 us: through Python autograd the host needs 130-170 us to issue
    # them (rocprofv3 shows a 44 us host gap between the forward and the backward kernel alone), i.e.
    # eager launches measure the host, at any N.  So the step's kernels — forward + backward — are
    # captured once into a hipGraph and replayed; the all-reduce is issued eagerly after the replay.
    # Falls back to eager launches if capture is unavailable.  The per-kernel HIP-event durations
    # (roofline) and `eager_ms_per_step` come from an eager pass right after the timed region.
    use_graph = args.graph in ('on', 'auto')
    graph = None
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            fence()
            graph = torch.cuda.CUDAGraph()
            x.grad = None
            with torch.cuda.graph(graph):
                d2 = man.pdist(x, squared=True, rows=(rb, re))
                grad_static, = torch.autograd.grad(d2, x, g_local)
            fence()
        except Exception as exc:  # noqa: BLE001 — report and measure eagerly instead
            if rank == 0:
                print(f'[bench] hipGraph capture unavailable ({type(exc).__name__}: {exc}); eager launches',
                      file=sys.stderr)
            graph = None
            fence()
    if graph is not None:
        # the collective stays OUTSIDE the captured graph (issued eagerly on the same stream right
        # after the replay): RCCL-in-graph capture cannot be validated on a 1-GPU box
        def run():
            graph.replay()
            if world > 1:
                dist.all_reduce(grad_static)
    else:
        run = step

    for _ in range(args.warmup):
        run()
    prof = (not args.no_prof) and graph is None   # event brackets cannot live inside a captured graph
    lib.call('mm_prof_enable', int(prof))
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    fence()
    elapsed = time.perf_counter() - t0
    lib.call('mm_prof_enable', 0)
    roofline_pass = 'timed region'
    eager_ms = None
    if graph is not None and not args.no_prof:
        # per-kernel durations: a short eager pass right after the timed region
        lib.call('mm_prof_enable', 1)
        k_eager = min(args.steps, 20)
        fence()
        te = time.perf_counter()
        for _ in range(k_eager):
            step()
        fence()
        eager_ms = (time.perf_counter() - te) / k_eager * 1e3
        lib.call('mm_prof_enable', 0)
        roofline_pass = 'eager pass after the (graph-replayed) timed region'

    kern = {}
    for name, kid in (('fwd', 0), ('bwd', 1)):
        cnt, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
        lib.call('mm_prof_collect', kid, ctypes.byref(cnt), ctypes.byref(ms))
        kern[name] = (ms.value / cnt.value * 1e-3) if cnt.value else None

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()

    if rank == 0:
        P = n * (n - 1) // 2
        pairs_local = hi - lo
        esz = 4
        out = {
            'metric': 'pairwise manifold-dist/sec (fwd+bwd), SPD(3) 5k-node',
            'value': P * args.steps / elapsed, 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'grqc-class graph, n={n} nodes -> SPD(3) affine-invariant, all '
                                   f'{P} pairs, squared distance + backward, reference init (||log X||=0.1)',
                       'pairs_per_step': P, 'parallelism': f'pair-rows sharded x{world}, 1 all-reduce',
                       'launch': 'hipGraph replay (fwd+bwd) + eager all-reduce' if graph is not None else 'eager'},
        }
        if eager_ms is not None:
            out['eager_ms_per_step'] = eager_ms   # same step issued through Python autograd (host-bound)
        if kern['bwd']:
            # dominant kernel: spd_pdist_bwd.  Algorithmic HBM bytes per launch: read g (4 B per
            # pair) + node factors in (2*6 floats) + accumulators out (2*6 floats) per node.
            by = pairs_local * esz + n * 24 * esz
            # VALU instructions per 64-pair wave iteration and HBM bytes per launch measured with
            # rocprofv3 --pmc on this exact workload (profiles/r01_v8_pmc_summary.txt); the plain
            # v_fma_f32 issue rate is 2.44 cycles per wave instruction per SIMD (tools/micro/valu_rate.hip)
            iters = pairs_local / 64.0
            issue_s = lambda instr, t: instr * iters * 2.44 / (1024 * 2.4e9) / t
            ref_shape = (n == N_NODES and world == 1)
            out['roofline'] = {'bound': 'hbm', 'kernel': 'spd_pdist_bwd_kernel<float,3,TI>',
                               'achieved': by / kern['bwd'] / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': by / kern['bwd'] / 1e9 / HBM_PEAK_GBS,
                               'traffic': 88.0e6 if ref_shape else None,
                               'traffic_source': 'rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, '
                                                 'profiles/r01_v8_pmc_summary.txt' if ref_shape else None,
                               'algorithmic_bytes': by,
                               'avg_launch_us': kern['bwd'] * 1e6, 'measured_in': roofline_pass,
                               'valu_issue_frac': issue_s(209, kern['bwd'])}
            if kern['fwd']:
                byf = pairs_local * esz + n * 12 * esz
                out['roofline_fwd'] = {'bound': 'hbm', 'kernel': 'spd_pdist_fwd_kernel<float,3,8>',
                                       'achieved': byf / kern['fwd'] / 1e9, 'peak': HBM_PEAK_GBS,
                                       'unit': 'GB/s', 'frac': byf / kern['fwd'] / 1e9 / HBM_PEAK_GBS,
                                       'traffic': 50.3e6 if ref_shape else None, 'algorithmic_bytes': byf,
                                       'avg_launch_us': kern['fwd'] * 1e6,
                                       'valu_issue_frac': issue_s(91, kern['fwd'])}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(42)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
