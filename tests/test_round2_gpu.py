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
        ('tools/fuzz_graph.py', ['30', '20269'])]


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
