#!/usr/bin/env python3
"""Secondary measurements: fwd+bwd time of the hot path on every BASELINE.json configuration
(synthetic inputs of the named sizes).  Prints one JSON object.
Usage: python tools/bench_configs.py [--only NAME]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from graphembed import manifolds as M  # noqa: E402
from graphembed._backend import unit_seed  # noqa: E402
from graphembed.modules import ManifoldEmbedding  # noqa: E402
from graphembed.objectives import StressLoss  # noqa: E402
from graphembed.optim import RiemannianAdam, RiemannianSGD  # noqa: E402


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def pdist_case(man, n, dtype, **kw):
    torch.manual_seed(42)
    x = man.rand(n, out=torch.empty(0, dtype=dtype, device='cuda'), **kw).requires_grad_()
    P = n * (n - 1) // 2
    g = torch.randn(P, dtype=dtype, device='cuda')
    fwd = timeit(lambda: man.pdist(x, squared=True))

    def both():
        d2 = man.pdist(x, squared=True)
        torch.autograd.grad(d2, x, g)
    tot = timeit(both)
    return {'n': n, 'pairs': P, 'dtype': str(dtype).split('.')[-1], 'fwd_us': fwd, 'fwd_bwd_us': tot,
            'pairs_per_s': P / (tot * 1e-6), 'GBps_8B_per_pair': P * 2 * x.element_size() / (tot * 1e-6) / 1e9}


def native_case(mans, n, dtype, loss='stress'):
    """full training step through ONE C-ABI call (mm_train_step_run / NativeTrainStep), replayed as a hipGraph — for a
    single SPD factor two launches per step: pair kernel + fused finalize / update / tables (bench.TrainStepWorkload)"""
    import bench
    wl = bench.TrainStepWorkload(mans, n, dtype, torch.device('cuda', 0), loss=loss)
    graph, _ = bench.graph_of(wl.kernels, bench.Fence(1))
    t = timeit(graph.replay, iters=40, warm=200)
    P = n * (n - 1) // 2
    return {'n': n, 'pairs': P, 'dtype': str(dtype).split('.')[-1], 'step_us': t, 'pairs_per_s': P / (t * 1e-6)}


def step_case(mans, n, dtype, fused=False, graph=False, pair_kernel=True, adam=False):
    """full training step: compute_dists + stress loss + backward + fused RSGD (momentum 0)"""
    torch.manual_seed(0)
    torch.set_default_dtype(dtype)
    try:
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, mans)
            emb.pair_kernel = pair_kernel  # products: the single mixed-manifold pair kernel
    finally:
        torch.set_default_dtype(torch.float32)
    P = n * (n - 1) // 2
    target = torch.rand(P, dtype=dtype, device='cuda') * 0.99 + 0.01
    fn = StressLoss()
    if adam:  # the optimizer of the paper grid (experiments/run_grid.py:24-36)
        opt = RiemannianAdam(list(emb.xs), lr=1e-3, exact=True, max_grad_norm=20)
        opt_s = RiemannianAdam(list(emb.scales), lr=1e-4, max_grad_norm=500)
    else:
        opt = RiemannianSGD(list(emb.xs), lr=1e-3, exact=True, max_grad_norm=20)
        opt_s = RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)

    def step():
        opt.zero_grad(set_to_none=True)
        opt_s.zero_grad(set_to_none=True)
        loss = emb.fused_objective(fn, target, None) if fused else fn(target, emb.compute_dists(None))
        loss.backward(unit_seed(loss))
        opt.step()
        opt_s.step()
    if graph:  # whole training step replayed as one hipGraph (static shapes, no host sync inside)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        t = timeit(gr.replay)
    else:
        t = timeit(step)
    return {'n': n, 'pairs': P, 'dtype': str(dtype).split('.')[-1], 'step_us': t, 'pairs_per_s': P / (t * 1e-6)}


CASES = {
    'c1_tree40_euclidean10_f64': lambda: pdist_case(M.Euclidean(10), 40, torch.float64),
    'c2_facebook_lorentz11_f32_gram': lambda: pdist_case(M.Lorentz(11), 4039, torch.float32),
    'c2_facebook_lorentz11_f32_valu': lambda: pdist_case(_valu(M.Lorentz(11)), 4039, torch.float32),
    'c2_facebook_lorentz11_f64_gram': lambda: pdist_case(M.Lorentz(11), 4039, torch.float64),
    'c2_facebook_lorentz11_f64_valu': lambda: pdist_case(_valu(M.Lorentz(11)), 4039, torch.float64),
    'c2_facebook_lorentz11_step_f32': lambda: step_case([M.Lorentz(11)], 4039, torch.float32),
    'c2_facebook_lorentz11_step_f32_fused': lambda: step_case([M.Lorentz(11)], 4039, torch.float32, fused=True),
    'c2_facebook_lorentz11_step_f32_fused_graph': lambda: step_case([M.Lorentz(11)], 4039, torch.float32, fused=True, graph=True),
    'c3_grqc_spd3_n5000_f32': lambda: pdist_case(M.SymmetricPositiveDefinite(3), 5000, torch.float32),
    'c3_grqc_spd3_n4158_f32': lambda: pdist_case(M.SymmetricPositiveDefinite(3), 4158, torch.float32),
    'c3_grqc_spd3_n5000_f64': lambda: pdist_case(M.SymmetricPositiveDefinite(3), 5000, torch.float64),
    # "mid-training" spread: ||log X|| = 0.35 -> pair distances ~0.5 (targets are normalised to max 1)
    'c3_grqc_spd3_n5000_f32_mid': lambda: pdist_case(M.SymmetricPositiveDefinite(3), 5000, torch.float32, ir=0.35),
    'c3_grqc_spd3_n5000_f64_mid': lambda: pdist_case(M.SymmetricPositiveDefinite(3), 5000, torch.float64, ir=0.35),
    'c3_spd3_stein_n5000_f32': lambda: pdist_case(M.SymmetricPositiveDefinite(3, use_stein_div=True), 5000, torch.float32),
    'c3_spd2_n5000_f32': lambda: pdist_case(M.SymmetricPositiveDefinite(2), 5000, torch.float32),
    'c4_csphd_product_step_f32': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float32),
    'c4_csphd_product_step_f32_fused': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float32, fused=True),
    'c4_csphd_product_step_f32_fused_graph': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float32, fused=True, graph=True),
    'c4_csphd_product_step_f32_fused_graph_unroll4': lambda: unrolled_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float32, 4),
    'c2_facebook_lorentz11_step_f32_fused_graph_unroll4': lambda: unrolled_case([M.Lorentz(11)], 4039, torch.float32, 4),
    'c4_csphd_product_step_f32_perfactor': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float32, fused=True, pair_kernel=False),
    'c4_csphd_product_step_f32_perfactor_graph': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float32, fused=True, graph=True, pair_kernel=False),
    'c4_csphd_product_step_f64_fused_graph': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float64, fused=True, graph=True),
    'c4_product_n5000_step_f32_fused_graph': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 5000, torch.float32, fused=True, graph=True),
    'c4_product_n5000_step_f32_perfactor_graph': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 5000, torch.float32, fused=True, graph=True, pair_kernel=False),
    'c4_product_n2500_step_f32_fused_graph': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 2500, torch.float32, fused=True, graph=True),
    'c4_product_n2500_step_f32_perfactor_graph': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 2500, torch.float32, fused=True, graph=True, pair_kernel=False),
    'c4_csphd_product_step_f32_fused_radam_graph': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float32, fused=True, graph=True, adam=True),
    'c3_spd3_step_n5000_f32_fused_radam_graph': lambda: step_case([M.SymmetricPositiveDefinite(3)], 5000, torch.float32, fused=True, graph=True, adam=True),
    'c3_spd3_step_n5000_f32_fused_radam': lambda: step_case([M.SymmetricPositiveDefinite(3)], 5000, torch.float32, fused=True, adam=True),
    'c4_csphd_product_step_f32_graph': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float32, graph=True),
    'c4_csphd_product_step_f64': lambda: step_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float64),
    'c5_wormnet_spd4_n2274_f32': lambda: pdist_case(M.SymmetricPositiveDefinite(4), 2274, torch.float32),
    'c5_wormnet_spd4_n16384_f32': lambda: pdist_case(M.SymmetricPositiveDefinite(4), 16384, torch.float32),
    'c3_spd3_step_n5000_f32': lambda: step_case([M.SymmetricPositiveDefinite(3)], 5000, torch.float32),
    'c3_spd3_step_n5000_f32_fused': lambda: step_case([M.SymmetricPositiveDefinite(3)], 5000, torch.float32, fused=True),
    'c3_spd3_step_n5000_f32_fused_graph': lambda: step_case([M.SymmetricPositiveDefinite(3)], 5000, torch.float32, fused=True, graph=True),
    'c3_spd3_step_n5000_f32_graph': lambda: step_case([M.SymmetricPositiveDefinite(3)], 5000, torch.float32, graph=True),
    'c3_spd3_step_n5000_f32_native_graph': lambda: native_case([M.SymmetricPositiveDefinite(3)], 5000, torch.float32),
    'c3_spd3_step_n5000_f64_native_graph': lambda: native_case([M.SymmetricPositiveDefinite(3)], 5000, torch.float64),
    'c4_csphd_product_step_f32_native_graph': lambda: native_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float32),
    'c2_facebook_lorentz11_step_f32_native_graph': lambda: native_case([M.Lorentz(11)], 4039, torch.float32),
    'c5_spd4_step_n16384_f32_native_graph': lambda: native_case([M.SymmetricPositiveDefinite(4)], 16384, torch.float32, loss='quotient'),
    'c5_spd4_step_n2274_f32_native_graph': lambda: native_case([M.SymmetricPositiveDefinite(4)], 2274, torch.float32, loss='quotient'),
    'c5_spd4_step_n2274_f32_fused': lambda: step_case([M.SymmetricPositiveDefinite(4)], 2274, torch.float32, fused=True),
    'c3_spd3_minibatch512_step_f32': lambda: minibatch_case([M.SymmetricPositiveDefinite(3)], 5000, 512, torch.float32),
    'c3_spd3_minibatch512_step_f32_graph': lambda: minibatch_case([M.SymmetricPositiveDefinite(3)], 5000, 512, torch.float32, graph=True),
    'c4_csphd_minibatch512_step_f32_graph': lambda: minibatch_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, 512, torch.float32, graph=True),
    'c4_csphd_minibatch512_step_f32_radam_graph': lambda: minibatch_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, 512, torch.float32, graph=True, adam=True),
    'c3_spd3_minibatch512_step_f32_radam_graph': lambda: minibatch_case([M.SymmetricPositiveDefinite(3)], 5000, 512, torch.float32, graph=True, adam=True),
    'c4_csphd_minibatch512_step_f32': lambda: minibatch_case([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, 512, torch.float32),
    'c2_lorentz11_minibatch512_step_f32': lambda: minibatch_case([M.Lorentz(11)], 4039, 512, torch.float32),
    # round 4: node minibatches inside the single factors' own pair kernels (SPD(4...9), vectors wider than 16)
    'c5_spd4_minibatch512_step_n16384_f32_graph': lambda: minibatch_case([M.SymmetricPositiveDefinite(4)], 16384, 512, torch.float32, graph=True),
    'c5_spd4_minibatch512_step_n16384_f32_native_graph': lambda: native_minibatch_case([M.SymmetricPositiveDefinite(4)], 16384, 512, torch.float32),
    'c5_spd4_minibatch512_step_n2274_f32_native_graph': lambda: native_minibatch_case([M.SymmetricPositiveDefinite(4)], 2274, 512, torch.float32),
    'c5_spd4_minibatch512_step_n2274_f32_radam_native_graph': lambda: native_minibatch_case([M.SymmetricPositiveDefinite(4)], 2274, 512, torch.float32, adam=True),
    'c3_spd3_minibatch512_step_f32_native_graph': lambda: native_minibatch_case([M.SymmetricPositiveDefinite(3)], 5000, 512, torch.float32),
    'spd6_minibatch512_step_n2274_f32_native_graph': lambda: native_minibatch_case([M.SymmetricPositiveDefinite(6)], 2274, 512, torch.float32),
    'lorentz24_minibatch512_step_n4039_f32_native_graph': lambda: native_minibatch_case([M.Lorentz(24)], 4039, 512, torch.float32),
    'sphere6_n5000_f32': lambda: pdist_case(M.Sphere(6), 5000, torch.float32),
    'euclidean10_n5000_f32': lambda: pdist_case(M.Euclidean(10), 5000, torch.float32),
    'grassmann52_n2000_f32': lambda: pdist_case(M.Grassmann(5, 2), 2000, torch.float32),
}


def unrolled_case(mans, n, dtype, unroll):
    """full-batch fused training step, `unroll` steps per recorded graph (GraphedTrainStep(unroll=)); time per STEP"""
    from graphembed.graphed import GraphedTrainStep
    torch.manual_seed(0)
    torch.set_default_dtype(dtype)
    try:
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, mans)
    finally:
        torch.set_default_dtype(torch.float32)
    P = n * (n - 1) // 2
    target = torch.rand(P, dtype=dtype, device='cuda') * 0.99 + 0.01
    fn = StressLoss()
    opts = [RiemannianSGD(list(emb.xs), lr=1e-3, exact=True, max_grad_norm=20),
            RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)]
    step = GraphedTrainStep(lambda: emb.fused_objective(fn, target, None), opts, warmup=2, unroll=unroll).capture()
    t = timeit(step) / unroll
    return {'n': n, 'pairs': P, 'dtype': str(dtype).split('.')[-1], 'step_us': t, 'pairs_per_s': P / (t * 1e-6)}


def minibatch_case(mans, n, bs, dtype, graph=False, adam=False):
    """node-minibatch training step as train.py:198-222 runs it (batch_size=512 in the paper grid): targets of the
    induced sub-graph from the dense matrix, embedding rows gathered, fused loss, scatter-add backward, RSGD"""
    from graphembed.data import GraphDataset
    from graphembed.modules import BatchedObjective
    torch.manual_seed(0)
    torch.set_default_dtype(dtype)
    try:
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, mans)
            ds = GraphDataset(torch.rand(n * (n - 1) // 2) * 0.99 + 0.01)
    finally:
        torch.set_default_dtype(torch.float32)
    obj = BatchedObjective(StressLoss(), ds, emb)
    if adam:
        opt = RiemannianAdam(list(emb.xs), lr=1e-3, exact=True, max_grad_norm=20)
        opt_s = RiemannianAdam(list(emb.scales), lr=1e-4, max_grad_norm=500)
    else:
        opt = RiemannianSGD(list(emb.xs), lr=1e-3, exact=True, max_grad_norm=20)
        opt_s = RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)
    perm = torch.randperm(n, device='cuda')
    state = {'i': 0}

    def step():
        i = state['i']
        idx = perm[i:i + bs]
        state['i'] = (i + bs) % (n - bs)
        opt.zero_grad()
        opt_s.zero_grad()
        obj(idx).backward()
        opt.step()
        opt_s.step()
    if graph:  # static index buffer, refreshed in place before every replay
        from graphembed.graphed import GraphedTrainStep
        idx_static = perm[:bs].clone()
        gstep = GraphedTrainStep(lambda: obj(idx_static), [opt, opt_s]).capture()

        def step():  # noqa: F811
            i = state['i']
            idx_static.copy_(perm[i:i + bs])
            state['i'] = (i + bs) % (n - bs)
            gstep()
    t = timeit(step, iters=50)
    P = bs * (bs - 1) // 2
    return {'n': n, 'pairs': P, 'dtype': str(dtype).split('.')[-1], 'step_us': t, 'pairs_per_s': P / (t * 1e-6)}


def native_minibatch_case(mans, n, bs, dtype, adam=False):
    """node-minibatch training step through ONE C-ABI call (mm_train_step_run with batch_idx), replayed as a hipGraph with the
    index vector refreshed in place: for a single SPD(d <= 5) factor two launches per step — the pair kernel over the batch (index
    vector inside) and the per-point finalize / optimizer / tables kernel over all n points (the launch count is read off the rocprofv3 kernel trace of this case: tools/gpu_r04_b.sh)."""
    from graphembed.data import GraphDataset
    from graphembed.native_step import NativeTrainStep
    torch.manual_seed(0)
    torch.set_default_dtype(dtype)
    try:
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, mans)
            ds = GraphDataset(torch.rand(n * (n - 1) // 2) * 0.99 + 0.01)
    finally:
        torch.set_default_dtype(torch.float32)
    if adam:
        opts = [RiemannianAdam(list(emb.xs), lr=1e-3, exact=True, max_grad_norm=20), RiemannianAdam(list(emb.scales), lr=1e-4, max_grad_norm=500)]
    else:
        opts = [RiemannianSGD(list(emb.xs), lr=1e-3, exact=True, max_grad_norm=20), RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)]
    step = NativeTrainStep(emb, StressLoss(), None, opts, dense=ds.pdists)
    perm = torch.randperm(n, device='cuda')
    idx_static = perm[:bs].clone()
    for _ in range(3):
        step(indices=idx_static)            # (the second step on: the tables of the new points are the step kernel's own)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            step(indices=idx_static)
    torch.cuda.current_stream().wait_stream(side)
    state = {'i': 0}

    def run():
        i = state['i']
        idx_static.copy_(perm[i:i + bs])
        state['i'] = (i + bs) % (n - bs)
        g.replay()
    t = timeit(run, iters=50)
    P = bs * (bs - 1) // 2
    return {'n': n, 'pairs': P, 'dtype': str(dtype).split('.')[-1], 'step_us': t, 'pairs_per_s': P / (t * 1e-6)}


def _valu(man):
    man.use_gram = False
    return man


def host_calibration():
    """How fast THIS box's host dispatches work — the yardstick for every eager (non-graph) row, which is host-bound: the same
    tree measured 1.3 - 1.6 x apart in its eager rows on two boxes while every graph-replayed row agreed to 3 % (round 5's
    table against round 4's; advisor, round 5).  `launch_us`: a trivial in-place torch kernel enqueued in a loop (framework
    dispatch + hipLaunchKernel, no synchronisation inside); `cabi_us`: an argument-check-only C-ABI call through ctypes;
    `python_us`: 1000 iterations of a pure-python arithmetic loop."""
    import time
    from graphembed import _backend as B
    t = torch.zeros(64, device='cuda')
    for _ in range(200):
        t.add_(1.0)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(2000):
            t.add_(1.0)
        best = min(best, (time.perf_counter() - t0) / 2000 * 1e6)
        torch.cuda.synchronize()
    fn = B.lib().raw('mm_pair_offset')
    t0 = time.perf_counter()
    for _ in range(20000):
        fn(5000, 17)
    cabi = (time.perf_counter() - t0) / 20000 * 1e6
    t0 = time.perf_counter()
    acc = 0
    for i in range(200000):
        acc += i * i % 7
    py = (time.perf_counter() - t0) / 200 * 1e6
    return {'launch_us': best, 'cabi_us': cabi, 'python_us': py, 'host_cpus': os.cpu_count()}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    out = {'_host': host_calibration()}
    for name, fn in CASES.items():
        if a.only and a.only not in name:
            continue
        out[name] = fn()
    print(json.dumps(out, indent=1))
