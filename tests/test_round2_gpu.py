"""Round-2 GPU tests: workspace lifetime under graph replay, bench.py's N > 1 path as a 2-rank
dry run on one GPU, a seeded slice of every fuzz campaign."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_graph_replay_survives_eager_calls_of_other_shapes():
    """A captured training step bakes the pair-kernel workspace's address and its "clean" flag into the
    graph.  Eager fused_objective calls with OTHER shapes between replays (a full-batch validation loss, a
    shorter tail minibatch) must neither free that workspace nor hand an uninitialised one to a later call:
    the replayed trajectory equals the undisturbed eager one."""
    from graphembed import manifolds as M
    from graphembed.data import GraphDataset
    from graphembed.graphed import GraphedTrainStep
    from graphembed.modules import BatchedObjective, ManifoldEmbedding
    from graphembed.objectives import StressLoss
    from graphembed.optim import RiemannianSGD
    n, bs = 300, 96
    fn = StressLoss()
    torch.set_default_dtype(torch.float64)
    try:
        def build():
            torch.manual_seed(9)
            with torch.device('cuda'):
                emb = ManifoldEmbedding(n, [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)])
                with torch.no_grad():
                    emb.perturb(0.3)
                ds = GraphDataset(torch.rand(n * (n - 1) // 2) * 3 + 0.5)
            obj = BatchedObjective(fn, ds, emb)
            opts = [RiemannianSGD(list(emb.xs), lr=0.01, exact=True, max_grad_norm=20),
                    RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)]
            return emb, ds, obj, opts
        torch.manual_seed(1)
        perm = torch.randperm(n, device='cuda')
        batches = [perm[k * bs:(k + 1) * bs].clone() for k in range(3)] * 2
        tail = perm[:40].clone()

        emb_e, ds_e, obj_e, opts_e = build()
        for b in batches:
            for o in opts_e:
                o.zero_grad()
            obj_e(b).backward()
            for o in opts_e:
                o.step()

        emb_g, ds_g, obj_g, opts_g = build()
        idx_static = batches[0].clone()
        step = GraphedTrainStep(lambda: obj_g(idx_static), opts_g, warmup=1).capture()
        ws_keys = set(emb_g._pair_ws)
        for b in batches[1:]:
            # disturbances between replays: other shapes through the same embedding's workspace table
            with torch.no_grad():
                full = emb_g.fused_objective(fn, ds_g[None], None)         # n = 300: another key
                short = obj_g(tail)                                          # 40 nodes: a third key
                junk = [torch.full((1 << 18, ), float('nan'), device='cuda') for _ in range(4)]   # reuse freed memory
                del junk
            assert torch.isfinite(full) and torch.isfinite(short)
            idx_static.copy_(b)
            step()
        assert ws_keys <= set(emb_g._pair_ws), 'the workspace a graph refers to was dropped'
        assert len(emb_g._pair_ws) == 3
        for a, b in zip(emb_g.xs, emb_e.xs):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-8, atol=1e-10)
        # a shape first seen DURING capture is not marked clean by the recording alone
        emb_c, ds_c, obj_c, opts_c = build()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            obj_c(batches[0])          # warm-up with bs = 96
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            rec = obj_c(tail)          # first call with 40 nodes happens while capturing
        entry = [e for k, e in emb_c._pair_ws.items() if k[2] == 40][0]
        assert entry[1] is False and entry[2] is True
        eager = obj_c(tail)            # must not trust the never-executed workspace
        g.replay()
        torch.cuda.synchronize()
        assert abs(eager.item() - rec.item()) <= 1e-10 * abs(eager.item())
    finally:
        torch.set_default_dtype(torch.float32)


def test_bench_two_ranks_on_one_gpu_gloo():
    """`python bench.py --gpus 2` starts its two ranks itself; with MM_BENCH_BACKEND=gloo they share the
    GPU (a dry run of the N > 1 path: sharded rows, graph replay, all-reduce, MAX-reduced time, per-rank
    phases, the config-5 block).  The JSON is kept under gpurun_out/ for profiles/."""
    env = dict(os.environ, MM_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '10', '--warmup', '3',
                        '--no-cpu-baseline', '--launch-timeout', '600'], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 10 and out['value'] > 0
    assert [p['rank'] for p in out['per_rank']] == [0, 1]
    assert sum(p['pairs'] for p in out['per_rank']) == 5000 * 4999 // 2
    for p in out['per_rank']:
        assert p['kernels_us'] > 0 and p['allreduce_us'] > 0 and p['host_gap_us'] >= 0
    cfg5 = [e for e in out['extra'] if 'config 5' in e['workload']]
    assert len(cfg5) == 1 and cfg5[0]['n_gpus'] == 2 and len(cfg5[0]['per_rank']) == 2
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'bench_gloo2_dryrun.json'), 'w') as f:
        f.write(lines[0] + '\n')


def test_bench_rank_watchdog_exits_nonzero():
    """A rank that outlives --rank-timeout exits 3 on its own."""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '5000000', '--warmup', '1',
                        '--no-cpu-baseline', '--no-extra', '--rank-timeout', '20'], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert 'rank-timeout' in r.stderr


# A fixed-seed slice of every randomised campaign (the campaigns themselves, thousands of cases each, are run by
# hand: tests/fuzz_*.py, tools/fuzz_*.py).  Each script exits non-zero on the first failing case.
FUZZ = [('tests/fuzz_pdist.py', ['50', '20261']), ('tests/fuzz_pdist.py', ['6', '20262', '--big']),
        ('tests/fuzz_optim.py', ['50', '20263']), ('tests/fuzz_maps.py', ['50', '20264']),
        ('tests/fuzz_misc.py', ['50', '20265']), ('tests/fuzz_metrics.py', ['40', '20266']),
        ('tools/fuzz_product.py', ['50', '20267']), ('tools/fuzz_product.py', ['30', '20268', '--single']),
        ('tools/fuzz_product.py', ['8', '20270', '--big']),
        ('tools/fuzz_graph.py', ['30', '20269']), ('tools/fuzz_step.py', ['40', '20271']), ('tools/fuzz_step.py', ['4', '20272', '--big']),
        ('tools/fuzz_walk.py', ['60', '20273'])]   # (round 5: shares cut on the host, closed-form block search, block-entry costs)


@pytest.mark.parametrize('script,argv', FUZZ, ids=[f'{os.path.basename(s)[:-3]}{"-" + a[-1][2:] if a[-1].startswith("--") else ""}'
                                                    for s, a in FUZZ])
def test_fuzz_slice(script, argv, monkeypatch, capsys):
    import importlib.util
    path = os.path.join(ROOT, script)
    spec = importlib.util.spec_from_file_location('fuzz_' + os.path.basename(script)[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    monkeypatch.setattr(sys, 'argv', [path] + argv)
    dtype = torch.get_default_dtype()
    try:
        spec.loader.exec_module(mod)
        try:
            mod.main()
        except SystemExit as e:
            assert not e.code, f'{script} {argv}: {capsys.readouterr().out[-2000:]}'
    finally:
        torch.set_default_dtype(dtype)
    assert 'ok' in capsys.readouterr().out


# ------------------------------------------------------------------ one C-ABI call per training step
@pytest.mark.parametrize('case', ['spd3_rsgd', 'lorentz11_adam', 'product_momentum', 'spd4_adam_f32'])
def test_native_train_step_matches_eager_loop(case):
    """`NativeTrainStep` (mm_train_step_run: objective + optimizer kernels issued from C++) advances the parameters,
    the optimizer state and the losses exactly like the eager loop of train.py:198-222 on the same classes."""
    import copy
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import QuotientLoss, StressLoss
    from graphembed.optim import RiemannianAdam, RiemannianSGD
    dt = torch.float32 if case.endswith('f32') else torch.float64
    mk = {'spd3_rsgd': lambda: [M.SymmetricPositiveDefinite(3)], 'lorentz11_adam': lambda: [M.Lorentz(11)],
          'product_momentum': lambda: [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)],
          'spd4_adam_f32': lambda: [M.SymmetricPositiveDefinite(4)]}[case]
    n = 150
    torch.set_default_dtype(dt)
    try:
        torch.manual_seed(3)
        with torch.device('cuda'):
            emb_a = ManifoldEmbedding(n, mk())
            with torch.no_grad():
                emb_a.perturb(0.3)
            target = torch.rand(n * (n - 1) // 2) * 0.9 + 0.05
        emb_b = copy.deepcopy(emb_a)
        fn = QuotientLoss() if 'adam' in case else StressLoss()

        def opts(emb):
            if 'adam' in case:
                return [RiemannianAdam(list(emb.xs), lr=1e-2, exact=True, max_grad_norm=20),
                        RiemannianAdam(list(emb.scales), lr=1e-3, max_grad_norm=500)]
            mom = 0.9 if 'momentum' in case else 0
            return [RiemannianSGD(list(emb.xs), lr=0.01, momentum=mom, dampening=0.1 if mom else 0, exact=True, max_grad_norm=20),
                    RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)]
        oa, ob = opts(emb_a), opts(emb_b)
        la, lb = [], []
        for epoch in range(6):               # eager reference loop
            loss = emb_a.fused_objective(fn, target, None, epoch=epoch, alpha=1.0)
            for o in oa:
                o.zero_grad(set_to_none=True)
            loss.backward()
            for o in oa:
                o.step()
            la.append(loss.item())
        step = NativeTrainStep(emb_b, fn, target, ob)
        lib, calls = step and __import__('graphembed')._backend.lib(), []
        orig = lib.call
        lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            for epoch in range(6):
                lb.append(step(epoch=epoch, alpha=1.0).item())
        finally:
            del lib.call
        if 'momentum' not in case:
            assert calls == ['mm_train_step_run'] * 6, calls
        else:   # the first heavy-ball step creates the buffers through the optimizers
            assert calls.count('mm_train_step_run') == 5, calls
        tol = 1e-4 if dt == torch.float32 else 1e-10
        np.testing.assert_allclose(lb, la, rtol=tol)
        for a, b in zip(list(emb_a.xs) + list(emb_a.scales), list(emb_b.xs) + list(emb_b.scales)):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=tol * 10, atol=tol)
        # frozen scales (burn-in, modules.py:36-39): read by the objective, not stepped
        emb_b.burnin(True)
        before = [s.detach().clone() for s in emb_b.scales]
        step(epoch=6, alpha=1.0)
        for s, s0 in zip(emb_b.scales, before):
            assert torch.equal(s.detach(), s0)
    finally:
        torch.set_default_dtype(torch.float32)


def test_native_train_step_host_cost():
    """The point of the entry point: an eager (un-captured) training step whose host side is one call.  Wall time per
    step of the native step within 1.35x of the replayed graph's, where the Python-driven eager loop is several times
    slower (SPD(3), n = 2000: the kernels are ~25 us)."""
    import copy
    import time
    from graphembed import manifolds as M
    from graphembed.graphed import GraphedTrainStep
    from graphembed.modules import ManifoldEmbedding
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import StressLoss
    from graphembed.optim import RiemannianSGD
    n = 2000
    torch.manual_seed(0)
    with torch.device('cuda'):
        emb = ManifoldEmbedding(n, [M.SymmetricPositiveDefinite(3)])
        target = torch.rand(n * (n - 1) // 2) * 0.9 + 0.05
    fn = StressLoss()

    def opts(e):
        return [RiemannianSGD(list(e.xs), lr=1e-3, exact=True, max_grad_norm=20), RiemannianSGD(list(e.scales), lr=1e-4, max_grad_norm=500)]

    def timed(f, k=200):
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e6
    e1, e2, e3 = emb, copy.deepcopy(emb), copy.deepcopy(emb)
    o1 = opts(e1)

    def eager():
        loss = e1.fused_objective(fn, target, None)
        for o in o1:
            o.zero_grad(set_to_none=True)
        loss.backward()
        for o in o1:
            o.step()
    t_eager = timed(eager)
    native = NativeTrainStep(e2, fn, target, opts(e2))
    t_native = timed(lambda: native())
    graphed = GraphedTrainStep(lambda: e3.fused_objective(fn, target, None), opts(e3)).capture()
    t_graph = timed(lambda: graphed())
    print(f'step wall us: eager {t_eager:.1f}  native {t_native:.1f}  graph replay {t_graph:.1f}')
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'native_step_host_cost.json'), 'w') as f:
        json.dump({'n': n, 'eager_us': t_eager, 'native_us': t_native, 'graph_us': t_graph}, f)
    assert t_native <= 1.35 * t_graph + 5.0, (t_native, t_graph)
    assert t_native < t_eager
