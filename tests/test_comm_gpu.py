"""The collective behind the C ABI (mm_comm_init / mm_allreduce_sum, csrc/comm.hip) and the sharded one-call training
step (mm_train_step with a row range and a communicator) — the replacement of the reference's only parallel call site,
torch.nn.DataParallel around BatchedObjective (graphembed/graphembed/train.py:107-109).

One MI355X per test box: RCCL is exercised with a ONE-RANK communicator (init, the all-reduce on the launch stream, its
capture into a HIP graph together with the kernels around it); the arithmetic of sharding is checked by running the
row ranges of 2 / 3 / 8 ranks one after the other on the same GPU and summing by hand what the all-reduce would sum."""
import copy
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def comm():
    from graphembed.comm import Communicator, available
    assert available(), 'librccl could not be bound'
    c = Communicator(0, 1, Communicator.unique_id(), torch.device('cuda', 0))
    yield c
    c.destroy()


def test_one_rank_communicator_allreduce_is_the_identity(comm):
    from graphembed import _backend as B
    assert comm.world == 1 and comm.rank == 0
    assert B.lib().raw('mm_comm_rccl_version')() >= 20000
    for dt in (torch.float32, torch.float64):
        x = torch.randn(100003, dtype=dt, device='cuda')
        want = x.clone()
        comm.all_reduce_(x)
        torch.cuda.synchronize()
        assert torch.equal(x, want)
    with pytest.raises(ValueError):
        comm.all_reduce_(torch.zeros(4, 4, device='cuda').t())
    with pytest.raises(B.BackendError):
        comm.all_reduce_(torch.zeros(4))


def test_allreduce_is_capturable_in_a_hip_graph(comm):
    """kernel -> mm_allreduce_sum -> kernel recorded as ONE graph and replayed (what bench.py does for N > 1)."""
    x = torch.arange(1024, dtype=torch.float32, device='cuda')
    y = torch.zeros_like(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.all_reduce_(x.clone())
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        t = x * 2
        comm.all_reduce_(t)
        y.copy_(t + 1)
    for k in range(3):
        x.add_(1.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, x * 2 + 1)


def _embedding(case, n, dt):
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    mk = {'spd3': lambda: [M.SymmetricPositiveDefinite(3)], 'spd4': lambda: [M.SymmetricPositiveDefinite(4)],
          'lorentz11': lambda: [M.Lorentz(11)],
          'product': lambda: [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)]}[case]
    torch.set_default_dtype(dt)
    try:
        torch.manual_seed(11)
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, mk())
            with torch.no_grad():
                emb.perturb(0.3)
            target = torch.rand(n * (n - 1) // 2) * 0.9 + 0.05
    finally:
        torch.set_default_dtype(torch.float32)
    return emb, target


def _optimizers(emb, adam):
    from graphembed.optim import RiemannianAdam, RiemannianSGD
    if adam:
        return [RiemannianAdam(list(emb.xs), lr=1e-2, exact=True, max_grad_norm=20),
                RiemannianAdam(list(emb.scales), lr=1e-3, max_grad_norm=500)]
    return [RiemannianSGD(list(emb.xs), lr=0.01, exact=True, max_grad_norm=20),
            RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)]


@pytest.mark.parametrize('case', ['spd3', 'spd4', 'lorentz11', 'product'])
@pytest.mark.parametrize('dt', [torch.float32, torch.float64], ids=['f32', 'f64'])
def test_sharded_step_with_one_rank_communicator_captured_equals_the_single_gpu_step(comm, case, dt):
    """mm_train_step_run with {row range of rank 0 of 1, communicator} — objective -> all-reduce -> optimizer — recorded
    as ONE HIP graph and replayed, against the plain single-GPU step issued eagerly.  Same kernels, same launch shapes;
    the gradient sums are float atomics, so two runs of the SAME step differ in the last bits: the bound is a few ulps
    of the step (it is bit-exact whenever the plain step reproduces itself)."""
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import QuotientLoss, StressLoss
    from graphembed.parallel import PairShard
    n = 190
    emb_a, target = _embedding(case, n, dt)
    emb_b, emb_c = copy.deepcopy(emb_a), copy.deepcopy(emb_a)
    adam = case in ('spd4', 'lorentz11')
    fn_a, fn_b, fn_c = (QuotientLoss(), QuotientLoss(), QuotientLoss()) if adam else (StressLoss(), StressLoss(), StressLoss())
    plain = NativeTrainStep(emb_a, fn_a, target, _optimizers(emb_a, adam))
    again = NativeTrainStep(emb_c, fn_c, target, _optimizers(emb_c, adam))
    shard = PairShard(n, world=1, rank=0)
    sharded = NativeTrainStep(emb_b, fn_b, target, _optimizers(emb_b, adam), shard=shard, comm=comm)
    assert sharded._desc.comm and sharded._desc.reduce_count == sharded.flat.numel()
    if adam:
        for f in (fn_a, fn_b, fn_c):
            f.on_device('cuda')
    kw = dict(epoch=2, alpha=1.0)
    sharded(**kw), plain(**kw), again(**kw)          # one uncaptured step each (RCCL warm-up, optimizer state)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
        loss_b = sharded(**kw)
    steps = 4
    la, lb, lc = [], [], []
    for _ in range(steps):
        la.append(plain(**kw).item())
        lc.append(again(**kw).item())
        graph.replay()
        lb.append(loss_b.item())
    eps = 1.2e-7 if dt == torch.float32 else 2.3e-16
    noise = max(abs(a - c) / abs(a) for a, c in zip(la, lc))
    np.testing.assert_allclose(lb, la, rtol=max(64 * eps, 4 * noise))
    for a, b, c in zip(list(emb_a.xs) + list(emb_a.scales), list(emb_b.xs) + list(emb_b.scales),
                       list(emb_c.xs) + list(emb_c.scales)):
        a, b, c = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy(), c.detach().double().cpu().numpy()
        self_noise = np.abs(a - c).max()
        assert np.abs(a - b).max() <= max(4 * self_noise, 256 * eps * np.abs(a).max()), (np.abs(a - b).max(), self_noise)


def _raw_step(emb, fn_spec, target_slice, rows, flat, ws, frozen=True):
    """One mm_train_step_run through ctypes with every parameter frozen (MM_OPT_NONE): objective + gradients only."""
    from graphembed import _backend as B
    from graphembed.native_step import _TrainStep, _factor_of, OPT_NONE
    xs, scales = list(emb.xs), list(emb.scales)
    k, n = len(xs), xs[0].shape[0]
    d = _TrainStep()
    d.dtype, d.n, d.nf = B.dtype_code(xs[0]), n, k
    d.loss_kind = B.LOSS_STRESS if fn_spec[0] == 'stress' else B.LOSS_QUOTIENT
    d.alpha, d.eps, d.terms = fn_spec[1], fn_spec[2], fn_spec[3]
    d.wmin, d.wmax = 1e-8, 1e8
    sizes = [x.numel() for x in xs]
    parts = torch.split(flat, sizes + [1 + k])
    for i, x in enumerate(xs):
        q = d.points[i]
        q.kind, q.dim = _factor_of(emb.manifolds[i])
        q.count, q.x, q.grad, q.optimizer = n, x.data_ptr(), parts[i].data_ptr(), OPT_NONE
        s = d.scales[i]
        s.kind, s.dim, s.count, s.x, s.optimizer = B.EUCLIDEAN, 1, 1, scales[i].data_ptr(), OPT_NONE
    d.target, d.loss_out, d.ws = target_slice.data_ptr(), parts[k].data_ptr(), ws.data_ptr()
    d.row_begin, d.row_end = rows
    B.lib().call('mm_train_step_run', ctypes.byref(d), B.stream_of(flat))
    return d


@pytest.mark.parametrize('case', ['spd3', 'lorentz11', 'product'])
@pytest.mark.parametrize('world', [2, 3, 8])
def test_row_shards_of_the_one_call_step_sum_to_the_full_step(case, world):
    """What the all-reduce sums: the {gradients, loss, scale gradients} records of the ranks' row ranges add up to the
    record of the whole pair list (fp64: to rounding), frozen parameters are left untouched (MM_OPT_NONE), and a
    descriptor whose gradient buffers lie outside reduce_buf is refused."""
    from graphembed import _backend as B
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import StressLoss
    n = 173
    emb, target = _embedding(case, n, torch.float64)
    before = [p.detach().clone() for p in list(emb.xs) + list(emb.scales)]
    probe = NativeTrainStep(emb, StressLoss(), target, _optimizers(emb, False))   # (allocates a workspace of the right size)
    spec = StressLoss().fused_spec()
    total = torch.zeros_like(probe.flat)
    full = torch.zeros_like(probe.flat)
    ws = torch.zeros_like(probe.ws)
    _raw_step(emb, spec, target, (0, n), full, ws)
    for r in range(world):
        rb, re = B.shard_rows(n, world, r)
        part = torch.zeros_like(probe.flat)
        ws = torch.zeros_like(probe.ws)
        assert 0 <= rb < re <= n
        _raw_step(emb, spec, target[B.pair_offset(n, rb):B.pair_offset(n, re)].contiguous(), (rb, re), part, ws)
        total += part
    torch.cuda.synchronize()
    np.testing.assert_allclose(total.cpu().numpy(), full.cpu().numpy(), rtol=1e-11, atol=1e-12 * float(full.abs().max()))
    for p, b in zip(list(emb.xs) + list(emb.scales), before):
        assert torch.equal(p.detach(), b)


def test_sharded_descriptor_argument_checks(comm):
    from graphembed import _backend as B
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import StressLoss
    from graphembed.parallel import PairShard
    n = 64
    emb, target = _embedding('spd3', n, torch.float32)
    step = NativeTrainStep(emb, StressLoss(), target, _optimizers(emb, False), shard=PairShard(n, world=1, rank=0), comm=comm)
    step()
    d = step._desc
    rc = B.lib().raw('mm_train_step_run')
    keep = d.reduce_count
    d.reduce_count = 8                      # the gradient buffer no longer lies inside the message
    assert rc(ctypes.byref(d), None) == -1
    d.reduce_count = keep
    d.row_begin, d.row_end = 10, 5
    assert rc(ctypes.byref(d), None) == -1
    d.row_begin, d.row_end = 0, n + 1
    assert rc(ctypes.byref(d), None) == -1
    with pytest.raises(ValueError):
        NativeTrainStep(emb, StressLoss(), target, _optimizers(emb, False), comm=comm)   # a communicator needs its shard
    with pytest.raises(ValueError):   # several ranks, no communicator: every rank would step on its partial gradient
        NativeTrainStep(emb, StressLoss(), target, _optimizers(emb, False), shard=PairShard(n, world=2, rank=0))
