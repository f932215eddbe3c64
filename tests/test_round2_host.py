"""Host-side behaviour added in round 2 (no GPU): optimizer step hooks, minibatch index validation,
the device loss schedule's write-on-change, bench.py's self-launch / time-outs."""
import json
import os
import subprocess
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_optimizer_step_hooks_run():
    """torch's step wrapper is skipped (host time), the hooks it would dispatch are not: per-optimizer and
    global pre/post hooks run, in torch's order, around the fused step."""
    from graphembed.optim import RiemannianAdam, RiemannianSGD
    from torch.optim.optimizer import register_optimizer_step_post_hook, register_optimizer_step_pre_hook
    for cls in (RiemannianSGD, RiemannianAdam):
        p = torch.nn.Parameter(torch.randn(4, 3))
        opt = cls([p], lr=0.1)
        seen = []
        h1 = opt.register_step_pre_hook(lambda o, a, k: seen.append('pre'))
        h2 = opt.register_step_post_hook(lambda o, a, k: seen.append('post'))
        g1 = register_optimizer_step_pre_hook(lambda o, a, k: seen.append('gpre'))
        g2 = register_optimizer_step_post_hook(lambda o, a, k: seen.append('gpost'))
        try:
            p.grad = torch.randn_like(p)
            before = p.detach().clone()
            opt.step()
            assert seen == ['gpre', 'pre', 'post', 'gpost'], seen
            assert not torch.equal(before, p.detach())
        finally:
            for h in (h1, h2, g1, g2):
                h.remove()
        seen.clear()
        p.grad = torch.randn_like(p)
        opt.step()
        assert seen == []


def test_minibatch_index_validation():
    """The in-kernel minibatch path needs distinct in-range indices: host-side index tensors are checked —
    repeats fall back to the gather / scatter path, out-of-range raises like the reference's x[i]."""
    from graphembed.modules import BatchedObjective
    bo = BatchedObjective.__new__(BatchedObjective)
    assert bo._distinct_in_range(torch.randperm(50)[:20], 50)
    assert not bo._distinct_in_range(torch.tensor([1, 2, 2, 5]), 50)
    assert not bo._distinct_in_range(torch.tensor([-1, 2, 3]), 50)   # negative = python-style, not served in-kernel
    with pytest.raises(IndexError):
        bo._distinct_in_range(torch.tensor([1, 50]), 50)
    with pytest.raises(IndexError):
        bo._distinct_in_range(torch.tensor([-51, 3]), 50)
    bo.check_indices = False
    assert bo._distinct_in_range(torch.tensor([1, 1]), 50)
    # the module-level forms every in-kernel route goes through (BatchedObjective, fused_objective, NativeTrainStep)
    from graphembed.modules import distinct_in_range, normalise_indices
    assert distinct_in_range(torch.randperm(50)[:20], 50) and not distinct_in_range(torch.tensor([4, 4]), 50)
    assert distinct_in_range(torch.empty(0, dtype=torch.int64), 50)
    assert normalise_indices(None, 50) is None
    assert normalise_indices(torch.tensor([-1, 2, -50]), 50).tolist() == [49, 2, 0]
    keep = torch.tensor([5, 6])
    assert normalise_indices(keep, 50) is keep
    for bad in ([1, 50], [-51, 3]):
        with pytest.raises(IndexError):
            normalise_indices(torch.tensor(bad), 50)


def test_index_scan_runs_once_per_step_and_the_opt_out_skips_it(monkeypatch):
    """Advisor (round 5): `check_indices = False` no longer disabled the ~20 us `aminmax` + `unique` on the in-kernel route
    and host-side indices were scanned up to four times per step.  Now: ONE `prepare_indices` per step at the entry point,
    every route below is told (`validated=True`), and the opt-out runs no scan at all."""
    import graphembed.modules as Mod
    from graphembed.modules import BatchedObjective, prepare_indices
    idx, ok = prepare_indices(torch.tensor([-1, 2, -50]), 50)
    assert idx.tolist() == [49, 2, 0] and ok
    assert prepare_indices(torch.tensor([4, -46]), 50)[1] is False            # a repeat after wrapping
    assert prepare_indices(torch.tensor([4, 4]), 50, distinct=False)[1] is True
    assert prepare_indices(None, 50) == (None, True)
    for bad in ([1, 50], [-51, 3]):
        with pytest.raises(IndexError):
            prepare_indices(torch.tensor(bad), 50)
    calls = {'aminmax': 0, 'unique': 0, 'minmax': 0}
    real_aminmax, real_unique = torch.aminmax, torch.unique
    monkeypatch.setattr(torch, 'aminmax', lambda *a, **k: (calls.__setitem__('aminmax', calls['aminmax'] + 1), real_aminmax(*a, **k))[1])
    monkeypatch.setattr(torch, 'unique', lambda *a, **k: (calls.__setitem__('unique', calls['unique'] + 1), real_unique(*a, **k))[1])
    seen = []

    class Emb:                       # (records what the routes are told; no kernels involved)
        n = 50
        xs = [torch.zeros(50, 3)]
        device = torch.device('cpu')

        def __len__(self):
            return 50

        def fused_objective(self, fn, gd, i, validated=False, **k):
            seen.append(('fused', validated))
            # what the real method does with an unvalidated batch
            if i is not None and not validated:
                Mod.prepare_indices(i, 50)
            return None

        def compute_dists(self, i, validated=False):
            seen.append(('dists', validated))
            return Mod.take_rows(self.xs[0], i, validated=validated).sum(-1)

    class DS:
        def __getitem__(self, i):
            return torch.zeros(i.numel())

    bo = BatchedObjective(lambda g, d: (g - d).sum(), DS(), Emb())
    batch = torch.randperm(50)[:20]
    bo(batch)
    assert calls == {'aminmax': 1, 'unique': 1, 'minmax': 0} and seen == [('fused', True), ('dists', True)]
    bo.check_indices = False
    calls.update(aminmax=0, unique=0)
    bo(batch)
    assert calls['aminmax'] == 0 and calls['unique'] == 0
    # a direct caller of the embedding's methods still gets the scan (once)
    calls.update(aminmax=0, unique=0)
    Emb().compute_dists(batch)
    assert calls['unique'] == 0


def test_quotient_schedule_written_only_on_change(monkeypatch):
    from graphembed.objectives import QuotientLoss
    q = QuotientLoss()
    dyn = q.on_device('cpu')
    q.set_epoch(0, 1.0)                      # == the initial values: nothing to write
    assert dyn.tolist() == [1.0, 1.0]
    q.set_epoch(3, 0.5)
    assert dyn.tolist() == [0.5, 0.25]
    before = dyn.clone()
    dyn.zero_()                              # a write would restore it; an unchanged schedule must not write
    q.set_epoch(3, 0.5)
    assert dyn.tolist() == [0.0, 0.0]
    q.set_epoch(4, 0.5)
    assert dyn.tolist() == [0.5, 0.2] and before.tolist() == [0.5, 0.25]
    spec = q.fused_spec(epoch=9, alpha=2.0)
    assert spec[:4] == ('quotient', 2.0, 0.1, 3) and spec[4] is dyn and dyn.tolist() == [2.0, 0.1]


def test_loss_schedule_must_live_on_the_embedding_device():
    from graphembed import _backend as B
    x = torch.zeros(3)
    assert B.dyn_ptr(None, x) is None
    with pytest.raises(B.BackendError):
        B.dyn_ptr(torch.zeros(2, dtype=torch.float32), x)     # wrong dtype
    assert B.dyn_ptr(torch.zeros(2, dtype=torch.float64), x).value


def _bench(*argv, env=None, timeout=300):
    e = dict(os.environ)
    e.pop('RANK', None)
    e.pop('WORLD_SIZE', None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv], env=e, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_self_launch_fails_fast_without_gpus():
    """`python bench.py --gpus 2` with no launcher around it starts its ranks itself; with no GPU the ranks
    refuse (there is no CPU path) and the launcher hands the non-zero exit code on — no hang."""
    if torch.cuda.is_available():
        pytest.skip('needs a GPU-less host')
    t0 = time.time()
    r = _bench('--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--launch-timeout', '240')
    assert r.returncode != 0
    assert 'needs an MI355X' in (r.stderr + r.stdout)
    assert time.time() - t0 < 240


def test_bench_launch_timeout_kills_the_job(tmp_path):
    """A job that outlives --launch-timeout is killed as a process group and the exit code is 124."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # the launcher with `torch.distributed.run` replaced by a sleeper: same Popen / wait / killpg path
    marker = tmp_path / 'pid'
    sleeper = tmp_path / 'sleep.py'
    sleeper.write_text(f'import os, time\nopen({str(marker)!r}, "w").write(str(os.getpid()))\ntime.sleep(600)\n')
    code = (
        'import sys, subprocess\n'
        f'sys.path.insert(0, {ROOT!r})\n'
        'import bench\n'
        'real = subprocess.Popen\n'
        f'subprocess.Popen = lambda cmd, **kw: real([sys.executable, {str(sleeper)!r}], **kw)\n'
        "args = bench.parse_args(['--gpus', '2', '--launch-timeout', '2'])\n"
        "bench.launch(args, ['--gpus', '2'])\n")
    t0 = time.time()
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 124, (r.returncode, r.stderr)
    assert time.time() - t0 < 60
    pid = int(marker.read_text())
    time.sleep(0.5)
    with pytest.raises(ProcessLookupError):
        os.kill(pid, 0)


def test_pmc_stamp_goes_stale_with_the_sources(tmp_path, monkeypatch):
    """bench.py quotes PMC traffic only from a profiles/pmc_head.json whose source hash matches the tree."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    os.makedirs(tmp_path / 'profiles')
    os.makedirs(tmp_path / 'matrix-manifolds_amd' / 'csrc')
    (tmp_path / 'matrix-manifolds_amd' / 'csrc' / 'k.hip').write_text('// v1\n')
    h1 = bench.kernel_source_hash()
    (tmp_path / 'profiles' / 'pmc_head.json').write_text(json.dumps({'kernel_source_hash': h1, 'kernels': {}}))
    assert bench.stamped_pmc() is not None
    (tmp_path / 'matrix-manifolds_amd' / 'csrc' / 'k.hip').write_text('// v2\n')
    assert bench.stamped_pmc() is None


def test_spd_entry_points_validate_arguments_before_touching_the_gpu():
    """Argument errors come back as MM_ERR_ARG from the C ABI without a launch (no GPU needed): null pointers, row
    ranges outside [0, n], more nodes than the 32-bit table offsets address (include/mm_manifolds.h, mm_spd_pdist_fwd)."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, 'matrix-manifolds_amd', 'lib', 'libmm_manifolds.so'))
    f = lib.mm_spd_pdist_fwd
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_double,
                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    fake = C.c_void_p(4096)          # never dereferenced: every call below is rejected first
    MM_F32, MM_ERR_ARG = 0, -1
    assert f(MM_F32, fake, (1 << 22) + 1, 3, 0, 1, 1, 1e-8, 1e8, fake, fake, 0, None) == MM_ERR_ARG
    assert f(MM_F32, None, 10, 3, 0, 10, 1, 1e-8, 1e8, fake, fake, 0, None) == MM_ERR_ARG
    assert f(MM_F32, fake, 10, 3, 0, 11, 1, 1e-8, 1e8, fake, fake, 0, None) == MM_ERR_ARG
    assert f(MM_F32, fake, 10, 3, 5, 4, 1, 1e-8, 1e8, fake, fake, 0, None) == MM_ERR_ARG
