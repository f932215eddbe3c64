"""The fused SPD training step of mm_train_step_run (csrc/spd.hip, spd_fused_step_kernel): pair kernel -> ONE per-point
kernel that finishes the gradient (spd_pdist_finalize_kernel's arithmetic), applies the optimizer rule
(optim/rsgd.py:29-82, optim/radam.py:62-98), writes the new point AND its per-node tables (spd_prep_kernel's arithmetic:
the next step passes MM_WS_PREPARED) and updates a momentum-free RSGD scale.  Checked against the eager loop on the same
classes (which is pinned to the reference's golden RSGD / RAdam traces in test_spd_gpu.py / test_radam.py), for every
rule, d = 2..5, both dtypes, and with the tables invalidated from outside in the middle of a run."""
import copy
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


def _setup(d, n, dt, spread=0.3):
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    torch.set_default_dtype(dt)
    try:
        torch.manual_seed(5)
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, [M.SymmetricPositiveDefinite(d)])
            with torch.no_grad():
                emb.perturb(spread)
            target = torch.rand(n * (n - 1) // 2) * 0.9 + 0.05
    finally:
        torch.set_default_dtype(torch.float32)
    return emb, target


def _opts(emb, rule, scale_rule='rsgd'):
    from graphembed.optim import RiemannianAdam, RiemannianSGD
    # (learning rates small enough that six steps stay a descent: a diverging run amplifies fp32 rounding chaotically)
    pts = {'rsgd': lambda p: RiemannianSGD(p, lr=2e-3, exact=True, max_grad_norm=20),
           'rsgd_retr': lambda p: RiemannianSGD(p, lr=2e-3, exact=False, max_grad_norm=None),
           'momentum': lambda p: RiemannianSGD(p, lr=1e-3, momentum=0.5, dampening=0.1, exact=True, max_grad_norm=20),
           'adam': lambda p: RiemannianAdam(p, lr=1e-2, exact=True, max_grad_norm=20),
           'adam_nc': lambda p: RiemannianAdam(p, lr=1e-2, betas=(0.9, None), nc=True, exact=False, max_grad_norm=None)}[rule]
    sc = {'rsgd': lambda p: RiemannianSGD(p, lr=1e-4, max_grad_norm=500),
          'rsgd_noclip': lambda p: RiemannianSGD(p, lr=1e-4, max_grad_norm=None),
          'momentum': lambda p: RiemannianSGD(p, lr=1e-5, momentum=0.5, max_grad_norm=500),
          'adam': lambda p: RiemannianAdam(p, lr=1e-3, max_grad_norm=500)}[scale_rule]
    return [pts(list(emb.xs)), sc(list(emb.scales))]


def _eager(emb, fn, target, opts, epochs):
    out = []
    for epoch in range(epochs):
        loss = emb.fused_objective(fn, target, None, epoch=epoch, alpha=1.0)
        for o in opts:
            o.zero_grad(set_to_none=True)
        loss.backward()
        for o in opts:
            o.step()
        out.append(loss.item())
    return out


def _close(a, b, dt, what):
    """fp64: element by element to rounding.  fp32: relative to the largest entry — the two runs differ in the order of the
    pair kernel's float atomics, and several optimizer steps (Adam's division by the root of a small second moment most of
    all) amplify that for single points: SPD(5) / Adam showed 4e-4 of max|x| on one point in two of six runs, while every
    fp64 case — the same code — agrees to 1e-9."""
    if dt == torch.float32:
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        ref = max(float(np.abs(a).max()), 1e-30)
        assert float(np.abs(a - b).max()) <= 2e-3 * ref, f'{what}: {float(np.abs(a - b).max()) / ref:.3e} of the largest entry'
        return
    np.testing.assert_allclose(b, a, rtol=1e-9, atol=1e-11, err_msg=what)


@pytest.mark.parametrize('d', [2, 3, 4, 5])
@pytest.mark.parametrize('rule,scale_rule', [('rsgd', 'rsgd'), ('rsgd_retr', 'rsgd_noclip'), ('momentum', 'rsgd'),
                                             ('adam', 'adam'), ('adam_nc', 'momentum'), ('rsgd', 'adam')])
@pytest.mark.parametrize('dt', [torch.float32, torch.float64], ids=['f32', 'f64'])
def test_fused_spd_step_matches_the_eager_loop(d, rule, scale_rule, dt):
    from graphembed import _backend as B
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import QuotientLoss, StressLoss
    n, epochs = 131, 6
    emb_a, target = _setup(d, n, dt)
    emb_b = copy.deepcopy(emb_a)
    fn = QuotientLoss() if 'adam' in rule else StressLoss()
    la = _eager(emb_a, fn, target, _opts(emb_a, rule, scale_rule), epochs)
    ob = _opts(emb_b, rule, scale_rule)
    step = NativeTrainStep(emb_b, fn, target, ob)
    flags, lb = [], []
    for epoch in range(epochs):
        lb.append(step(epoch=epoch, alpha=1.0).item())
        flags.append(step._desc.ws_flags)
    first_fused = 1 if 'momentum' in (rule, scale_rule) else 0   # (the first heavy-ball step creates its buffer through the optimizer)
    assert flags[first_fused] == 0 and all(f == B.MM_WS_PREPARED for f in flags[first_fused + 1:]), flags
    _close(la, lb, dt, 'losses')
    for a, b in zip(list(emb_a.xs) + list(emb_a.scales), list(emb_b.xs) + list(emb_b.scales)):
        _close(a.detach().cpu().numpy(), b.detach().cpu().numpy(), dt, 'parameters')
    # the gradient left behind is the Euclidean gradient of the LAST step's loss at the points BEFORE its update
    assert torch.isfinite(emb_b.xs[0].grad).all() and emb_b.xs[0].grad.abs().max() > 0
    # optimizer state advanced alike
    oa = _opts(copy.deepcopy(emb_a), rule, scale_rule)   # (fresh: only the key set is compared below)
    sb = ob[0].state[emb_b.xs[0]]
    assert set(sb) >= ({'momentum_buffer'} if rule == 'momentum' else {'exp_avg', 'exp_avg_sq', 'step'} if 'adam' in rule else set())
    if 'adam' in rule:
        assert float(sb['step']) == epochs + 1


def test_tables_written_by_the_step_equal_a_fresh_preparation():
    """After a fused step the workspace holds exactly what mm_spd_prepare computes from the new points."""
    from graphembed import _backend as B
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import StressLoss
    for d, dt in ((3, torch.float32), (4, torch.float64)):
        n = 257
        emb, target = _setup(d, n, dt)
        step = NativeTrainStep(emb, StressLoss(), target, _opts(emb, 'rsgd'))
        step(), step()
        fresh = torch.zeros_like(step.ws)
        x = emb.xs[0]
        B.lib().call('mm_spd_prepare', B.dtype_code(x), B.ptr(x), n, d, B.ptr(fresh), B.stream_of(x))
        torch.cuda.synchronize()
        head = 64 + (n * 4 + 63) // 64 * 64           # status word + per-point flags (ints), then the tables (spd_ws.hpp)
        assert torch.equal(step.ws[:head], fresh[:head])
        # (behind the tables sits the share table of the balanced walk — 2048 entries of 32 bytes, 32-byte aligned: pair kernels
        # write it, a preparation does not)
        esz = x.element_size()
        end = head + (step.ws.numel() - 2048 * 32 - 31 - head) // esz * esz
        a, b = step.ws[head:end].view(dt), fresh[head:end].view(dt)
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-6 if dt == torch.float32 else 1e-14, atol=0)


def test_points_changed_from_outside_drop_the_prepared_flag():
    """An in-place edit of the points between two steps (stabilize / projx / a manual perturbation) invalidates the tables:
    the next step prepares again and matches the eager loop that did the same edit."""
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import StressLoss
    dt, n = torch.float64, 97
    emb_a, target = _setup(3, n, dt)
    emb_b = copy.deepcopy(emb_a)
    fn = StressLoss()
    oa, ob = _opts(emb_a, 'rsgd'), _opts(emb_b, 'rsgd')
    step = NativeTrainStep(emb_b, fn, target, ob)
    la, lb = [], []
    for epoch in range(5):
        if epoch == 3:
            with torch.no_grad():
                for e in (emb_a, emb_b):
                    e.xs[0].mul_(1.1)              # bumps the tensor's version counter
        la += _eager(emb_a, fn, target, oa, 1)
        lb.append(step().item())
        assert step._desc.ws_flags == (0 if epoch in (0, 3) else 1)
    np.testing.assert_allclose(lb, la, rtol=1e-10)


def test_wide_matrices_take_the_unfused_step():
    """SPD(6) is outside the fused step kernel's range (d <= 5): mm_train_step_run issues objective + finalize + update
    as separate launches and matches the eager loop all the same."""
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import StressLoss
    n, dt = 67, torch.float64
    emb_a, target = _setup(6, n, dt, spread=0.2)
    emb_b = copy.deepcopy(emb_a)
    la = _eager(emb_a, StressLoss(), target, _opts(emb_a, 'rsgd'), 3)
    step = NativeTrainStep(emb_b, StressLoss(), target, _opts(emb_b, 'rsgd'))
    lb = [step().item() for _ in range(3)]
    _close(la, lb, dt, 'losses')
    _close(emb_a.xs[0].detach().cpu().numpy(), emb_b.xs[0].detach().cpu().numpy(), dt, 'points')


def test_fused_step_frozen_points_take_the_unfused_objective():
    """MM_OPT_NONE on the points (no optimizer rule): the objective runs unfused and nothing is updated."""
    import ctypes
    from graphembed import _backend as B
    from graphembed.native_step import NativeTrainStep, OPT_NONE
    from graphembed.objectives import StressLoss
    emb, target = _setup(3, 64, torch.float32)
    step = NativeTrainStep(emb, StressLoss(), target, _opts(emb, 'rsgd'))
    step()
    before = emb.xs[0].detach().clone()
    d = step._desc
    d.points[0].optimizer = OPT_NONE
    d.scales[0].optimizer = OPT_NONE
    d.ws_flags = 0
    B.lib().call('mm_train_step_run', ctypes.byref(d), B.stream_of(step.target))
    torch.cuda.synchronize()
    assert torch.equal(emb.xs[0].detach(), before)
    loss = emb.fused_objective(StressLoss(), target, None)
    np.testing.assert_allclose(step.loss_out[0].item(), loss.item(), rtol=1e-5)


# ---- single vector factor (csrc/vec.hip, vec_fused_step_kernel; vec_step.hpp) ------------------------------------------------
def _vec_setup(man_name, m, n, dt, spread=0.3):
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    torch.set_default_dtype(dt)
    try:
        torch.manual_seed(7)
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, [getattr(M, man_name)(m)])
            with torch.no_grad():
                emb.perturb(spread)
            target = torch.rand(n * (n - 1) // 2) * 0.9 + 0.05
    finally:
        torch.set_default_dtype(torch.float32)
    return emb, target


@pytest.mark.parametrize('man_name,m', [('Lorentz', 11), ('Sphere', 6), ('Euclidean', 10), ('Lorentz', 3), ('Sphere', 16),
                                        ('Euclidean', 20)])
@pytest.mark.parametrize('rule,scale_rule', [('rsgd', 'rsgd'), ('rsgd_retr', 'rsgd_noclip'), ('momentum', 'rsgd'),
                                             ('adam', 'adam'), ('adam_nc', 'momentum'), ('rsgd', 'adam')])
@pytest.mark.parametrize('dt', [torch.float32, torch.float64], ids=['f32', 'f64'])
def test_fused_vector_step_matches_the_eager_loop(man_name, m, rule, scale_rule, dt):
    """Pair kernel (sums left in the workspace) + ONE per-point kernel: gradient, loss record, optimizer rule, padded copy
    of the new points, momentum-free RSGD scale — against the eager loop on the same classes (pinned to the reference's
    golden RSGD / RAdam traces in test_vec_gpu.py / test_radam.py)."""
    from graphembed import _backend as B
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import QuotientLoss, StressLoss
    assert B.lib().raw('mm_vec_fused_step_supports')(B.MM_F32 if dt == torch.float32 else B.MM_F64,
                                                     {'Euclidean': B.EUCLIDEAN, 'Lorentz': B.LORENTZ, 'Sphere': B.SPHERE}[man_name], m)
    n, epochs = 197, 6
    emb_a, target = _vec_setup(man_name, m, n, dt)
    emb_b = copy.deepcopy(emb_a)
    fn = QuotientLoss() if 'adam' in rule else StressLoss()
    la = _eager(emb_a, fn, target, _opts(emb_a, rule, scale_rule), epochs)
    ob = _opts(emb_b, rule, scale_rule)
    step = NativeTrainStep(emb_b, fn, target, ob)
    flags, lb = [], []
    for epoch in range(epochs):
        lb.append(step(epoch=epoch, alpha=1.0).item())
        flags.append(step._desc.ws_flags)
    first_fused = 1 if 'momentum' in (rule, scale_rule) else 0
    assert flags[first_fused] == 0 and all(f == B.MM_WS_PREPARED for f in flags[first_fused + 1:]), flags
    _close(la, lb, dt, 'losses')
    for a, b in zip(list(emb_a.xs) + list(emb_a.scales), list(emb_b.xs) + list(emb_b.scales)):
        _close(a.detach().cpu().numpy(), b.detach().cpu().numpy(), dt, 'parameters')
    g = emb_b.xs[0].grad
    assert torch.isfinite(g).all() and g.abs().max() > 0
    # the gradient left behind = the Euclidean gradient of the last step's loss at the points before its update: redo that
    # step's objective on a copy of the eager run, one step short
    emb_c, _ = _vec_setup(man_name, m, n, dt)
    oc = _opts(emb_c, rule, scale_rule)
    _eager(emb_c, fn, target, oc, epochs - 1)
    loss = emb_c.fused_objective(fn, target, None, epoch=epochs - 1, alpha=1.0)
    gc, = torch.autograd.grad(loss, [emb_c.xs[0]])
    scale = gc.abs().max().item()     # (relative to the largest entry: small entries of an fp32 sum carry its rounding)
    # (two runs of several fp32 steps whose atomics land in different orders: the points themselves agree to ~1e-6)
    assert (gc - g).abs().max().item() <= (2e-4 if dt == torch.float32 else 1e-9) * scale, 'gradient left in x.grad'
    # the workspace is prepared for the next step: the padded copy equals the points, the sums are clear
    pad = next(p for p in (4, 8, 12, 16, 24, 32) if p >= m)
    ws = step.ws.view(dt)
    acc, slots = n * (pad + 1), 2 * 256
    assert not ws[:acc + slots].any()
    xpad = ws[acc + slots:acc + slots + (n + 1) * pad].view(n + 1, pad)
    assert torch.equal(xpad[:n, :m], emb_b.xs[0].detach().view(n, m)) and not xpad[:, m:].any() and not xpad[n].any()


def test_vector_points_changed_from_outside_drop_the_prepared_flag():
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import StressLoss
    dt, n = torch.float64, 97
    emb_a, target = _vec_setup('Lorentz', 6, n, dt)
    emb_b = copy.deepcopy(emb_a)
    fn = StressLoss()
    oa, ob = _opts(emb_a, 'rsgd'), _opts(emb_b, 'rsgd')
    step = NativeTrainStep(emb_b, fn, target, ob)
    la, lb = [], []
    for epoch in range(5):
        if epoch == 3:
            with torch.no_grad():
                for e in (emb_a, emb_b):
                    e.xs[0].copy_(e.manifolds[0].projx(e.xs[0] * 1.01))   # bumps the tensor's version counter
        la += _eager(emb_a, fn, target, oa, 1)
        lb.append(step().item())
        assert step._desc.ws_flags == (0 if epoch in (0, 3) else 1)
    np.testing.assert_allclose(lb, la, rtol=1e-10)


def test_vector_step_outside_the_fused_range_is_unfused():
    """Lorentz(24) in fp32: the matrix-core objective, separate update launches — same result as the eager loop."""
    from graphembed import _backend as B
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import StressLoss
    assert not B.lib().raw('mm_vec_fused_step_supports')(B.MM_F32, B.LORENTZ, 24)
    emb_a, target = _vec_setup('Lorentz', 24, 150, torch.float32)
    emb_b = copy.deepcopy(emb_a)
    la = _eager(emb_a, StressLoss(), target, _opts(emb_a, 'rsgd'), 4)
    step = NativeTrainStep(emb_b, StressLoss(), target, _opts(emb_b, 'rsgd'))
    lb = [step().item() for _ in range(4)]
    assert step._desc.ws_flags == 0
    _close(la, lb, torch.float32, 'losses')
    _close(emb_a.xs[0].detach().cpu().numpy(), emb_b.xs[0].detach().cpu().numpy(), torch.float32, 'points')


# ---- product embeddings (csrc/product_pairs.hip, product_step_kernel; product_step.hpp) ----------------------------------------
def _product_setup(spec, n, dt, spread=0.3):
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    torch.set_default_dtype(dt)
    try:
        torch.manual_seed(11)
        with torch.device('cuda'):
            mans = [M.SymmetricPositiveDefinite(d) if name == 'SPD' else getattr(M, name)(d) for name, d in spec]
            emb = ManifoldEmbedding(n, mans)
            with torch.no_grad():
                emb.perturb(spread)
            target = torch.rand(n * (n - 1) // 2) * 0.9 + 0.05
    finally:
        torch.set_default_dtype(torch.float32)
    return emb, target


@pytest.mark.parametrize('spec,n', [((('Lorentz', 6), ('Sphere', 6), ('SPD', 2)), 131),      # BASELINE config 4's product
                                    ((('Euclidean', 5), ('SPD', 3)), 77),
                                    ((('Lorentz', 4), ('Euclidean', 16), ('Sphere', 3)), 150),
                                    ((('Sphere', 6), ('SPD', 2)), 3)])                       # fewer points than loss blocks
@pytest.mark.parametrize('rule,scale_rule', [('rsgd', 'rsgd'), ('rsgd_retr', 'rsgd_noclip'), ('momentum', 'rsgd'),
                                             ('adam', 'adam'), ('adam_nc', 'momentum'), ('rsgd', 'adam')])
@pytest.mark.parametrize('dt', [torch.float32, torch.float64], ids=['f32', 'f64'])
def test_fused_product_step_matches_the_eager_loop(spec, n, rule, scale_rule, dt):
    """Mixed-manifold pair kernel + ONE kernel (gradients, loss record, every factor's optimizer rule, RSGD scales) against
    the eager loop on the same classes; the launches are counted through the library's own profiling hooks only
    indirectly — what is asserted here is the arithmetic: losses, parameters, optimizer state, the gradients left behind."""
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import QuotientLoss, StressLoss
    epochs = 6
    emb_a, target = _product_setup(spec, n, dt)
    emb_b = copy.deepcopy(emb_a)
    fn = QuotientLoss() if 'adam' in rule else StressLoss()
    oa = _opts(emb_a, rule, scale_rule)
    la = _eager(emb_a, fn, target, oa, epochs)
    ob = _opts(emb_b, rule, scale_rule)
    step = NativeTrainStep(emb_b, fn, target, ob)
    lb = [step(epoch=epoch, alpha=1.0).item() for epoch in range(epochs)]
    _close(la, lb, dt, 'losses')
    for a, b in zip(list(emb_a.xs) + list(emb_a.scales), list(emb_b.xs) + list(emb_b.scales)):
        _close(a.detach().cpu().numpy(), b.detach().cpu().numpy(), dt, 'parameters')
    for xa, xb in zip(emb_a.xs, emb_b.xs):      # optimizer state of every factor advanced alike
        sa, sb = oa[0].state[xa], ob[0].state[xb]
        for key in ('momentum_buffer', 'exp_avg', 'exp_avg_sq'):
            if key in sa:
                # (relative to the buffer's largest entry: small entries of an fp32 accumulation carry its rounding)
                ref = sa[key].abs().max().item()
                assert (sa[key] - sb[key]).abs().max().item() <= (2e-4 if dt == torch.float32 else 1e-9) * ref, key
        if 'step' in sa:
            assert float(sa['step']) == float(sb['step']) == epochs + 1
    # the gradients left behind: those of the last step's loss at the parameters before its update
    emb_c, _ = _product_setup(spec, n, dt)
    _eager(emb_c, fn, target, _opts(emb_c, rule, scale_rule), epochs - 1)
    loss = emb_c.fused_objective(fn, target, None, epoch=epochs - 1, alpha=1.0)
    gs = torch.autograd.grad(loss, list(emb_c.xs) + list(emb_c.scales))
    for i, (g, p) in enumerate(zip(gs, list(emb_b.xs) + list(emb_b.scales))):
        scale = max(g.abs().max().item(), 1e-30)
        # (a scale's gradient is one fp32 sum over all pairs of terms of both signs, evaluated after five fp32 steps of two
        # trajectories whose sums are ordered differently: 4e-3 — the Euclidean(5) x SPD(3) case measured 2.1e-3 with the
        # ordered kernel forced once the loss sums left through one transposing reduction; fp64 holds 1e-9 on every path)
        tol = (4e-3 if i >= len(emb_b.xs) else 2e-4) if dt == torch.float32 else 1e-9
        assert (g - p.grad.view_as(g)).abs().max().item() <= tol * scale, (i, g, p.grad)
    # the accumulators and loss slots are left clean (MM_WS_CLEAN); behind them sits the symmetric pair kernel's node table
    es = 4 if dt == torch.float32 else 8
    acc = (1 + len(spec)) * 256 + sum(d * d * n if name == 'SPD' else 17 * n for name, d in spec)
    assert not step.ws[64:64 + es * acc].any()
