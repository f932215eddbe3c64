"""CPU-only tests: the C ABI loads and exports what include/mm_manifolds.h declares, the
pair-list geometry, loud failure without a GPU, and the host logic (RSGD control flow,
embedding containers, losses, pair-range sharding with its single all-reduce over gloo)."""
import ctypes
import itertools
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, sym

from graphembed import _backend as B


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'mm_manifolds.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return set(re.findall(r'\b(mm_[a-z0-9_]+)\s*\(', src))


def test_library_exports_every_declared_symbol():
    lib = B.lib()                       # raises if the .so is missing
    declared = header_symbols()
    assert len(declared) >= 25
    raw = ctypes.CDLL(lib.path)
    for name in declared:
        assert hasattr(raw, name), f'{name} declared in mm_manifolds.h but not exported'
    assert declared == set(B.SIGNATURES), declared ^ set(B.SIGNATURES)
    assert lib.raw('mm_target_arch')() == b'gfx950'
    assert lib.raw('mm_abi_version')() >= 2
    assert lib.raw('mm_spd_max_dim')() >= 4 and lib.raw('mm_vec_max_dim')() >= 16


def test_argument_errors_need_no_gpu():
    lib = B.lib()
    f = lib.raw('mm_spd_pdist_fwd')
    assert f(B.MM_F32, None, 10, 3, 0, 10, 1, 1e-8, 1e8, None, None, 0, None) == -1      # null x
    g = lib.raw('mm_vec_pdist_fwd')
    assert g(B.MM_F32, B.LORENTZ, None, 10, 11, 0, 10, 1, None, None) == -1
    buf = (ctypes.c_float * 4)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert f(B.MM_F32, p, 10, 3, 5, 4, 1, 1e-8, 1e8, p, p, 0, None) == -1                # row_begin > row_end
    assert g(B.MM_F32, B.LORENTZ, p, 10, 1000, 0, 10, 1, p, None) == -2                   # m too large
    with pytest.raises(B.BackendError):
        lib.call('mm_spd_pdist_fwd', B.MM_F32, None, 10, 3, 0, 10, 1, 1e-8, 1e8, None, None, 0, None)
    # (+ one padding row of nodeLC; round 6: + the share table — 2048 entries of 32 bytes behind the tables, 32-byte aligned)
    up32 = lambda b: (b + 31) // 32 * 32
    assert lib.raw('mm_spd_pdist_ws_bytes')(B.MM_F32, 5000, 3) == up32(64 + 20032 + 4 * (5000 * (6 * 6 + 9 + 1) + 512 + 2 * 6)) + 2048 * 32
    assert lib.raw('mm_spd_pdist_ws_bytes')(B.MM_F64, 7, 4) == up32(64 + 64 + 8 * (7 * (6 * 10 + 16 + 1) + 512 + 2 * 10)) + 2048 * 32


@pytest.mark.parametrize('n,world', [(1, 1), (2, 2), (5, 8), (40, 3), (5000, 8), (16384, 8), (4039, 7)])
def test_shard_rows_cover_and_balance(n, world):
    lib = B.lib()
    P = n * (n - 1) // 2
    prev_end, sizes = 0, []
    for r in range(world):
        a, b = ctypes.c_int64(), ctypes.c_int64()
        lib.call('mm_shard_rows', n, world, r, ctypes.byref(a), ctypes.byref(b))
        assert (a.value, b.value) == B.shard_rows(n, world, r)
        assert a.value == prev_end and b.value >= a.value
        prev_end = b.value
        lo, hi = lib.raw('mm_pair_offset')(n, a.value), lib.raw('mm_pair_offset')(n, b.value)
        assert (lo, hi) == (B.pair_offset(n, a.value), B.pair_offset(n, b.value))
        sizes.append(hi - lo)
    assert prev_end == n and sum(sizes) == P
    if n >= 64 * world:
        # balanced by COST (a pair of a row of L pairs counts 1 + L / 400000: csrc/common.hip) to within a row or two ...
        K = 400000
        rows = [B.shard_rows(n, world, r) for r in range(world)]
        cost = [sum((n - 1 - i) * (K + n - 1 - i) for i in range(a, b)) for a, b in rows]
        assert max(cost) - min(cost) <= 2 * (n - 1) * (K + n - 1)
        # ... which keeps the pair counts within the model's range (n = 16384: the first rank gets 11 % fewer pairs than the last)
        assert max(sizes) <= min(sizes) * (1 + 1.2 * n / K) + 2 * n


def test_pair_offset_matches_triu_indices():
    n = 37
    m = torch.triu_indices(n, n, 1)
    for k in [0, 1, 35, 36, 300, n * (n - 1) // 2 - 1]:
        i, j = int(m[0, k]), int(m[1, k])
        assert B.pair_offset(n, i) + (j - i - 1) == k


def test_fails_loudly_without_gpu_or_library(tmp_path):
    from graphembed.manifolds import Lorentz, SymmetricPositiveDefinite
    with pytest.raises(B.BackendError, match='CPU tensor'):
        SymmetricPositiveDefinite(3).pdist(torch.eye(3).repeat(4, 1, 1))
    with pytest.raises(B.BackendError, match='CPU tensor'):
        Lorentz(4).pdist(torch.randn(5, 4))
    with pytest.raises(B.BackendError, match='CPU tensor'):
        with torch.no_grad():
            Lorentz(4).exp(torch.randn(5, 4), torch.randn(5, 4))
    with pytest.raises(B.BackendError, match='not found'):
        B.HipLibrary(str(tmp_path / 'libmm_manifolds.so'))
    with pytest.raises(TypeError):
        B.dtype_code(torch.zeros(1, dtype=torch.float16))


def test_utils_and_vec_maps():
    from graphembed import utils
    from graphembed.manifolds import SymmetricPositiveDefinite as SPD
    assert utils.nnm1d2_to_n(780) == 40 and utils.nnp1d2_to_n(6) == 3
    with pytest.raises(AssertionError):
        utils.nnm1d2_to_n(781)
    v = torch.arange(10.)
    sq = utils.squareform1(v)
    assert sq.shape == (5, 5) and torch.equal(sq, sq.T) and torch.equal(utils.squareform1(sq), v)
    assert torch.equal(sq[0, 1:], v[:4])                 # row-major upper triangle
    u = torch.randn(7, 6)
    U = SPD.from_vec(u)
    assert torch.allclose(U, U.transpose(-1, -2)) and torch.allclose(SPD.to_vec(U), u, atol=1e-6)
    assert torch.allclose((U * U).sum((-1, -2)), (u * u).sum(-1), atol=1e-5)   # Vec is an isometry
    m = utils.triu_mask(4, d=1)
    assert m.sum() == 6 and not m[2, 2] and m[0, 3]


@pytest.mark.parametrize('key,args', [('spd3', ('spd', 3)), ('lorentz6', ('lorentz', 6)), ('sphere6', ('sphere', 6))])
def test_rsgd_control_flow_matches_reference(key, args):
    """graphembed.optim.RiemannianSGD (clip, momentum + transport, exact/retr, set_) driven
    through CPU stand-in manifolds reproduces the reference optimizer's two-step traces."""
    import cpu_double
    from graphembed.modules import ManifoldParameter
    from graphembed.optim import RiemannianSGD
    G = load_golden(key)
    man = cpu_double.make(*args)
    base = 'f64/rsgd'
    for exact, clip, mom in itertools.product([0, 1], [0, 1], [0, 1]):
        tag = f'{base}/exact{exact}_clip{clip}_mom{mom}'
        p = ManifoldParameter(torch.from_numpy(G[f'{base}/x0']).clone(), manifold=man)
        opt = RiemannianSGD([p], lr=0.05, momentum=0.9 if mom else 0, dampening=0.1 if mom else 0,
                            max_grad_norm=2.0 if clip else None, exact=bool(exact))
        for step, gk in ((1, 'g1'), (2, 'g2')):
            p.grad = torch.from_numpy(G[f'{base}/{gk}']).clone()
            opt.step()
            np.testing.assert_allclose(p.data.numpy(), G[f'{tag}/x{step}'], rtol=1e-9, atol=1e-11)
        if mom:
            np.testing.assert_allclose(opt.state[p]['momentum_buffer'].numpy(), G[f'{tag}/buf2'], rtol=1e-9, atol=1e-11)


def test_flat_parameters_use_euclidean_fallback():
    """rsgd.py:7,56-59: plain Parameters (scales) are stepped as Euclidean(1) points."""
    from graphembed.optim import RiemannianSGD
    s = torch.nn.Parameter(torch.tensor(0.5, dtype=torch.float64))
    opt = RiemannianSGD([s], lr=0.1, max_grad_norm=2.0)
    s.grad = torch.tensor(10.0, dtype=torch.float64)
    opt.step()
    assert s.item() == pytest.approx(0.5 - 0.1 * 2.0)
    with pytest.raises(ValueError):
        RiemannianSGD([s], lr=0.1, momentum=-1)


def test_embedding_container_and_losses():
    import cpu_double
    from graphembed.modules import BatchedObjective, ManifoldEmbedding, ManifoldParameter
    from graphembed.objectives import QuotientLoss, StressLoss, Sum
    from graphembed.data import GraphDataset
    G = load_golden('callers')
    mans = [cpu_double.make('lorentz', 6), cpu_double.make('sphere', 6), cpu_double.make('spd', 2)]
    emb = ManifoldEmbedding(33, mans)
    assert list(emb.state_dict()) == ['xs.0', 'xs.1', 'xs.2', 'scales.0', 'scales.1', 'scales.2']
    assert all(isinstance(p, ManifoldParameter) and p.manifold is m for p, m in zip(emb.xs, mans))
    emb = emb.double()
    assert all(p.manifold is m for p, m in zip(emb.xs, mans))      # tag survives Module._apply
    base = 'f64/product/batch'
    idx = torch.from_numpy(G['f64/product/idx'])
    with torch.no_grad():
        for k in range(3):
            emb.xs[k].copy_(torch.from_numpy(G[f'{base}/x_{k}']))
            emb.scales[k].fill_(float(G[f'{base}/scales'][k]))
    md = emb.compute_dists(idx)
    np.testing.assert_allclose(md.detach().numpy(), G[f'{base}/d2'], rtol=1e-9)
    gd, mdl = torch.from_numpy(G['f64/loss/gd']), torch.from_numpy(G['f64/loss/md'])
    assert StressLoss()(gd, mdl).item() == pytest.approx(float(G['f64/loss/stress']), rel=1e-12)
    assert QuotientLoss()(gd, mdl, epoch=3, alpha=1.7).item() == pytest.approx(float(G['f64/loss/quotient']), rel=1e-12)
    assert str(Sum(StressLoss(), QuotientLoss())) == 'stress_loss__quotient_loss'
    with pytest.raises(ValueError):
        QuotientLoss(inc_l1=False, inc_l2=False)
    # GraphDataset: squared, max-normalised targets; subset indexing follows pdist order
    gp = torch.from_numpy(G['tree40/gpdists'])
    ds = GraphDataset(gp.clone())
    np.testing.assert_allclose(ds[None].numpy(), G['tree40/target'], rtol=1e-12)
    sub = torch.tensor([7, 3, 20, 11])
    dense = torch.zeros(40, 40, dtype=gp.dtype)
    iu = torch.triu_indices(40, 40, 1)
    dense[iu[0], iu[1]] = ds[None]
    dense = dense + dense.T
    expect = torch.stack([dense[sub[a], sub[b]] for a in range(4) for b in range(a + 1, 4)])
    assert torch.equal(ds[sub], expect) and len(ds) == 40
    obj = BatchedObjective(StressLoss(), GraphDataset(torch.rand(33 * 32 // 2, dtype=torch.float64) + 0.1), emb)
    assert obj(idx).ndim == 0


def _sharded_worker(rank, world, port, tmp, key):
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import cpu_double
        from graphembed.modules import ManifoldEmbedding
        from graphembed.objectives import StressLoss
        from graphembed.optim import RiemannianSGD
        from graphembed.parallel import PairShard, sharded_compute_dists
        torch.manual_seed(0)                               # identical replicas
        mans = [cpu_double.make('lorentz', 6), cpu_double.make('spd', 3)]
        emb = ManifoldEmbedding(41, mans).double()
        target = torch.rand(41 * 40 // 2, dtype=torch.float64)
        shard = PairShard(41)
        assert (shard.world, shard.rank) == (world, rank)
        opt = RiemannianSGD(list(emb.xs) + list(emb.scales), lr=0.01, max_grad_norm=20)
        for _ in range(2):
            opt.zero_grad()
            md = sharded_compute_dists(emb, shard)
            assert md.numel() == shard.num_pairs
            loss = StressLoss()(shard.slice(target), md)
            loss.backward()
            opt.step()
        torch.save({'xs': [x.data for x in emb.xs], 'scales': [s.data for s in emb.scales],
                    'grads': [p.grad for p in list(emb.xs) + list(emb.scales)]}, f'{tmp}/{key}_{rank}.pt')
    finally:
        dist.destroy_process_group()


def test_pair_sharding_over_gloo_matches_single_process(tmp_path):
    """world_size 2 on CPU: each rank evaluates only its row range of the pair list; one
    all-reduce leaves the full gradient on both ranks; replicas stay identical and equal to
    the unsharded run."""
    import socket
    import torch.multiprocessing as mp
    import cpu_double
    from graphembed.modules import ManifoldEmbedding
    from graphembed.objectives import StressLoss
    from graphembed.optim import RiemannianSGD
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path), 'w2'), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'w2_{r}.pt') for r in range(2))
    torch.manual_seed(0)
    emb = ManifoldEmbedding(41, [cpu_double.make('lorentz', 6), cpu_double.make('spd', 3)]).double()
    target = torch.rand(41 * 40 // 2, dtype=torch.float64)
    opt = RiemannianSGD(list(emb.xs) + list(emb.scales), lr=0.01, max_grad_norm=20)
    for _ in range(2):
        opt.zero_grad()
        StressLoss()(target, emb.compute_dists(None)).backward()
        opt.step()
    for k in range(2):
        assert torch.equal(r0['xs'][k], r1['xs'][k])                       # replicas identical
        np.testing.assert_allclose(r0['xs'][k].numpy(), emb.xs[k].data.numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r0['scales'][k].numpy(), emb.scales[k].data.numpy(), rtol=1e-10)
    ref_grads = [p.grad for p in list(emb.xs) + list(emb.scales)]
    for a, b in zip(r0['grads'], ref_grads):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-9, atol=1e-12)


def test_sharded_fused_objective_declines_on_cpu():
    """No CPU fallback behind the fused path: on CPU tensors it answers None and the caller composes the
    step from sharded_compute_dists (which then fails loudly in pdist without the HIP library / a GPU)."""
    import torch
    from graphembed.objectives import StressLoss
    from graphembed.parallel import PairShard, sharded_fused_objective

    class Emb:
        xs = [torch.zeros(6, 3, requires_grad=True)]
        scales = [torch.zeros((), requires_grad=True)]

        def fused_objective(self, *a, **k):
            return None if not self.xs[0].is_cuda else 1
    assert sharded_fused_objective(Emb(), StressLoss(), torch.zeros(15), PairShard(6, world=2, rank=0)) is None


def test_spd_from_vec_batch_equal_to_vector_length():
    """from_vec of a batch of d(d+1)/2-vectors whose batch size is d(d+1)/2 (looks like one square matrix to
    the reference's squareform0, utils.py:68-87 — SPD(2).rand(3) raises there): always vector -> matrix."""
    import torch
    from graphembed.manifolds import SymmetricPositiveDefinite as SPD
    for d in (2, 3, 4):
        k = d * (d + 1) // 2
        v = torch.arange(1.0, k * k + 1).reshape(k, k)
        x = SPD.from_vec(v)
        assert x.shape == (k, d, d) and torch.equal(x, x.transpose(1, 2))
        back = SPD.to_vec(x)
        assert back.shape == (k, k) and torch.allclose(back, v)


def test_embedding_deepcopy_keeps_manifolds():
    """copy.deepcopy(embedding) (the training engine snapshots the best embedding, train.py:116-121): the copies of
    the ManifoldParameters keep their manifold and their values, and are independent tensors."""
    import copy
    import torch
    from graphembed.modules import ManifoldParameter

    class Man:
        pass
    man = Man()
    p = ManifoldParameter(torch.arange(6.0).reshape(2, 3), manifold=man)
    holder = torch.nn.ParameterList([p])
    dup = copy.deepcopy(holder)
    assert isinstance(dup[0], ManifoldParameter) and dup[0].manifold is man and dup[0].requires_grad
    assert torch.equal(dup[0].data, p.data) and dup[0].data_ptr() != p.data_ptr()
    q = ManifoldParameter(torch.zeros(2), manifold=man, requires_grad=False)
    assert copy.deepcopy(q).requires_grad is False and copy.deepcopy(q).manifold is man


# ---- round 3: the collective behind the C ABI, the sharded one-call step -----------------------------------------
def test_comm_entry_points_reject_bad_arguments_without_a_gpu():
    lib = B.lib()
    assert lib.raw('mm_allreduce_sum')(None, B.MM_F32, None, 4, None) == -1          # no communicator
    handle = ctypes.c_void_p()
    token = (ctypes.c_char * 128)()
    assert lib.raw('mm_comm_init')(ctypes.byref(handle), 3, 2, token, 0) == -1       # rank >= world
    assert lib.raw('mm_comm_init')(ctypes.byref(handle), 0, 0, token, 0) == -1       # world < 1
    assert lib.raw('mm_comm_init')(None, 0, 1, token, 0) == -1
    assert lib.raw('mm_comm_unique_id')(None) == -1
    assert lib.raw('mm_comm_destroy')(None) == 0
    assert lib.raw('mm_comm_world')(None) == 0 and lib.raw('mm_comm_rank')(None) == -1
    assert lib.raw('mm_comm_available')() in (0, 1)
    assert isinstance(lib.raw('mm_comm_last_error')(), bytes)


def test_train_step_struct_layout_matches_the_header(tmp_path):
    """The ctypes mirror of mm_train_step / mm_step_param (graphembed/native_step.py) against the C compiler's layout."""
    import shutil
    import subprocess
    from graphembed.native_step import _StepParam, _TrainStep
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    fields_p = [f[0] for f in _StepParam._fields_]
    fields_t = [f[0] for f in _TrainStep._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "mm_manifolds.h"', 'int main(void) {',
           'printf("%zu %zu\\n", sizeof(mm_step_param), sizeof(mm_train_step));']
    for f in fields_p:
        src.append(f'printf("p {f} %zu\\n", offsetof(mm_step_param, {f}));')
    for f in fields_t:
        src.append(f'printf("t {f} %zu\\n", offsetof(mm_train_step, {f}));')
    src += ['return 0; }']
    c = tmp_path / 'layout.c'
    c.write_text('\n'.join(src))
    exe = str(tmp_path / 'layout')
    subprocess.run(['gcc', '-std=c11', '-I' + os.path.join(ROOT, 'include'), str(c), '-o', exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split('\n')
    sp, st = map(int, out[0].split())
    assert (sp, st) == (ctypes.sizeof(_StepParam), ctypes.sizeof(_TrainStep))
    for line in out[1:]:
        if not line:
            continue
        which, name, off = line.split()
        cls = _StepParam if which == 'p' else _TrainStep
        assert getattr(cls, name).offset == int(off), (which, name)


def test_pair_row_chunks_cover_the_pair_vector_in_order():
    """graphembed.manifolds.spd._pair_row_chunks (the narrow-window pdist, advisor round 5: triu_indices(n, n) and the gather of ALL
    pairs were materialised even for a small row shard): chunks of whole rows, in pair-vector order, equal to the slice of
    triu_indices the old route took — any row range, any budget."""
    import warnings
    from graphembed import _backend as B
    from graphembed.manifolds.spd import SymmetricPositiveDefinite, _pair_row_chunks
    for n in (2, 3, 17, 64):
        iu = torch.triu_indices(n, n, 1)
        for rb, re in ((0, n), (0, 1), (n // 3, 2 * n // 3), (n - 2, n), (n - 1, n), (5 % n, 5 % n)):
            lo0, hi0 = B.pair_offset(n, rb), B.pair_offset(n, re)
            for budget in (1, 7, 40, 10 ** 6):
                at, ii, jj = 0, [], []
                for i, j, lo, hi in _pair_row_chunks(n, rb, re, budget, torch.device('cpu')):
                    assert lo == at and hi - lo == i.numel() == j.numel()
                    rows_in_chunk = int(i.max()) - int(i.min()) + 1
                    assert hi - lo <= budget or rows_in_chunk == 1     # whole rows; only a single row may exceed the budget
                    at = hi
                    ii.append(i); jj.append(j)
                assert at == hi0 - lo0
                if at:
                    assert torch.equal(torch.cat(ii), iu[0, lo0:hi0]) and torch.equal(torch.cat(jj), iu[1, lo0:hi0])
    # constructing a manifold whose window takes that route says so, once per process
    SymmetricPositiveDefinite._warned_narrow = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        SymmetricPositiveDefinite(3, wmin=0.5, wmax=2.0)
        SymmetricPositiveDefinite(4, wmin=0.5, wmax=2.0)
        SymmetricPositiveDefinite(3)                       # the default window: nothing to say
    assert len([x for x in w if 'narrower than' in str(x.message)]) == 1


def test_walk_arithmetic_closed_form_matches_the_search(tmp_path):
    """The resident-grid kernels cut their shares on the host (WalkShares) and find a share's first column block in closed
    form (ColWalk::find_fast: an fp32 square root, then stepped either way against the exact prefix sums until exact)
    instead of a binary search behind two 64-bit divisions:
    tools/micro/walk_check.hip holds both against the exact integer forms for launches and shards up to n = 2^22 — the plain
    line and the line with block-entry costs (cross = 1 ... 1024, incl. the shipped 16 / 8) — and requires the
    stepping to need at most four steps."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    exe = str(tmp_path / 'walk_check')
    subprocess.run([hipcc, '-O1', '-std=c++17', '--offload-arch=gfx950', os.path.join(ROOT, 'tools', 'micro', 'walk_check.hip'), '-o', exe],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'bad = 0' in r.stdout, r.stdout[-2000:] + r.stderr[-500:]


def test_train_step_struct_is_versioned_by_its_size():
    """mm_train_step.struct_size (ABI 4; advisor, round 4: the struct grew at its tail without a version field, so a caller
    built against the shorter header was read past its end).  The size checks return before anything touches a GPU."""
    from graphembed import _backend as B
    from graphembed.native_step import _TrainStep
    lib = B.lib()
    assert lib.raw('mm_abi_version')() >= 4
    run = lib.raw('mm_train_step_run')
    d = _TrainStep()
    assert d.struct_size == ctypes.sizeof(_TrainStep) and _TrainStep.struct_size.offset == 0
    d.dtype, d.n, d.nf = B.dtype_code(torch.zeros(1)), 4, 9          # nf = 9: MM_ERR_ARG from the body, i.e. the size was accepted
    assert run(ctypes.byref(d), None) == -1
    # an unversioned struct, and what an ABI <= 3 caller has in these bytes ({dtype, loss_kind} = small integers)
    for bad in (0, 1, (2 << 32) | 1, ctypes.sizeof(_TrainStep) - 8, 64 * ctypes.sizeof(_TrainStep)):
        d.struct_size = bad
        assert run(ctypes.byref(d), None) == -1, bad

    # a caller built against a LONGER (future) header: accepted while the tail the library does not know is zero
    class Longer(ctypes.Structure):
        _fields_ = [('base', _TrainStep), ('future_field', ctypes.c_int64), ('future_ptr', ctypes.c_void_p)]
    e = Longer()
    e.base.struct_size = ctypes.sizeof(Longer)
    e.base.dtype, e.base.n, e.base.nf = d.dtype, 4, 9
    assert run(ctypes.byref(e), None) == -1                            # (again the body's nf check)
    e.future_field = 3
    assert run(ctypes.byref(e), None) == -2                            # MM_ERR_UNSUPPORTED: a feature this library lacks


def test_sync_grads_uses_an_installed_communicator():
    """graphembed.parallel routes the step's collective through the installed communicator object (the RCCL one on a
    GPU box) and falls back to torch.distributed without one."""
    from graphembed import parallel

    class Fake:
        world, rank = 2, 0
        calls = 0

        def all_reduce_(self, t):
            Fake.calls += 1
            return t.mul_(2)
    a = torch.ones(3, requires_grad=True)
    b = torch.ones(2, requires_grad=True)
    parallel.set_communicator(Fake())
    try:
        x, y = parallel.sync_grads(a, b)
        (x.sum() + 3 * y.sum()).backward()
    finally:
        parallel.set_communicator(None)
    assert Fake.calls == 1                                  # ONE collective for all parameters
    assert torch.equal(a.grad, torch.full((3,), 2.0)) and torch.equal(b.grad, torch.full((2,), 6.0))
    a.grad = None
    x, = parallel.sync_grads(a)                             # no communicator, no process group: identity
    x.sum().backward()
    assert torch.equal(a.grad, torch.ones(3))


def test_cpp_autograd_nodes_build_load_and_refuse_cpu_tensors():
    """csrc_torch/mm_autograd.cpp (the C++ autograd nodes of `pdist`): builds against this interpreter's torch with g++ alone,
    binds the C ABI at run time, and a CPU tensor is refused with the package's own error — there is no CPU path behind it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('_mm_autograd_build', os.path.join(ROOT, 'matrix-manifolds_amd', 'csrc_torch', 'build.py'))
    autograd_build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(autograd_build)
    path = autograd_build.build_if_stale()
    assert os.path.isfile(path) and autograd_build.is_current()   # (the stamp: torch version + source hash the binary was built from)
    from graphembed import _backend as B
    from graphembed import manifolds as M
    B._autograd = False     # (an earlier test may have looked the module up before it was built)
    ext = B.autograd_ext()
    assert ext is not None and {'init', 'spd_pdist', 'vec_pdist'} <= set(dir(ext))
    with pytest.raises(B.BackendError):
        M.SymmetricPositiveDefinite(3).pdist(torch.eye(3).repeat(4, 1, 1))
    with pytest.raises(B.BackendError):
        M.Lorentz(4).pdist(torch.randn(5, 4))
