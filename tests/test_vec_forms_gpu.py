"""The two backward forms of the vector-manifold pdist — the symmetric VALU kernel (csrc/vec_sym.hpp, the default up to
m = 16; Lorentz / sphere flushed straight into the gradient) and the matrix-core one (csrc/vec_gram.hip, the default for
fp32 17 <= m <= 32 and the fp32 squared Euclidean distance) — each against the fp64 oracle through the C ABI, whichever
the host-side routing prefers; and the suite's own pdist / fused-loss tests re-run with the non-default form forced."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = {'f32': torch.float32, 'f64': torch.float64}
GREL = {'f32': 5e-4, 'f64': 1e-9}   # relative to max|grad| of the call (tests/test_vec_gpu.py)


def oracle_grad(name, m, x64, g64, squared):
    from oracle import ref_port as rp
    port = rp.make(name, m)
    xr = x64.clone().requires_grad_()
    d = port.pdist(xr, squared=squared)
    gr, = torch.autograd.grad((d * g64).sum(), xr)
    return gr.numpy()


def backward(form, kind, x, g, rb, re, squared):
    from graphembed import _backend as B
    lib = B.lib()
    n, m = x.shape
    dt = B.dtype_code(x)
    grad = torch.full_like(x, float('nan'))
    if form == 'gram':
        rc = lib.raw('mm_vec_pdist_bwd_gram')(dt, kind, B.ptr(x), B.ptr(g), n, m, rb, re, int(squared), B.ptr(grad), B.stream_of(x))
    else:
        ws = torch.empty(lib.raw('mm_vec_pdist_ws_bytes')(dt, n, m), dtype=torch.uint8, device=x.device)
        rc = lib.raw('mm_vec_pdist_bwd')(dt, kind, B.ptr(x), B.ptr(g), n, m, rb, re, int(squared), B.ptr(grad), B.ptr(ws),
                                        B.stream_of(x))
    torch.cuda.synchronize()
    return rc, grad


@pytest.mark.parametrize('name,m,n', [('lorentz', 11, 1000), ('sphere', 6, 700), ('lorentz', 16, 513), ('lorentz', 3, 300),
                                      ('sphere', 2, 129), ('lorentz', 20, 400), ('sphere', 24, 321), ('lorentz', 32, 200),
                                      ('euclidean', 10, 515), ('euclidean', 31, 260)])
@pytest.mark.parametrize('dname', list(DT))
@pytest.mark.parametrize('form', ['valu', 'gram'])
def test_backward_form_vs_oracle(name, m, n, dname, form):
    from graphembed import _backend as B
    from oracle import ref_port as rp
    kind = {'lorentz': B.LORENTZ, 'sphere': B.SPHERE, 'euclidean': B.EUCLIDEAN}[name]
    gen = torch.Generator().manual_seed(n + m)
    x64 = rp.make(name, m).rand(n, ir=0.5 if name == 'sphere' else 0.3, dtype=torch.float64, generator=gen)
    g64 = torch.randn(n * (n - 1) // 2, dtype=torch.float64, generator=gen)
    x = x64.to(DT[dname]).cuda()
    for squared in (True, False):
        if dname == 'f32' and not squared:
            # d(acosh q)/dq and d(acos q)/dq are singular at q = 1: an fp32 ulp of the inner product of a close pair moves its
            # weight by percents in ANY fp32 evaluation (the reference's included — tests/test_vec_gpu.py check_grad), so the
            # plain distance is checked on the pairs further than 0.1 apart
            d = rp.make(name, m).pdist(x64, squared=False)
            g64 = torch.where(d < 0.1, torch.zeros_like(g64), g64)
        g = g64.to(DT[dname]).cuda()
        rc, grad = backward(form, kind, x, g, 0, n, squared)
        if rc == -2:   # MM_ERR_UNSUPPORTED
            # the matrix cores: fp32 up to m = 32 (fp64 16), Euclidean only squared and in fp32
            assert form == 'gram' and (m > (32 if dname == 'f32' else 16) or
                                       (name == 'euclidean' and (dname == 'f64' or not squared or m > 31)))
            continue
        assert rc == 0
        ref = oracle_grad(name, m, x64, g64, squared)
        err = np.abs(grad.double().cpu().numpy() - ref).max() / np.abs(ref).max()
        assert err <= GREL[dname], f'{form} {name}{m} squared={squared}: {err:.3e}'
        # row shards: every call clears and fills the whole gradient with its own rows' pairs — the parts sum to it
        total = torch.zeros_like(grad)
        for r in range(3):
            rb, re = B.shard_rows(n, 3, r)
            lo, hi = B.pair_offset(n, rb), B.pair_offset(n, re)
            gpart = torch.zeros_like(g)
            gpart[lo:hi] = g[lo:hi]
            rc, part = backward(form, kind, x, g[lo:hi].contiguous(), rb, re, squared)   # g: the shard's own pairs
            assert rc == 0
            rc, masked = backward(form, kind, x, gpart, 0, n, squared)   # the same pairs through a full-range call
            scale = masked.abs().max().item()
            assert (part - masked).abs().max().item() <= 0.2 * GREL[dname] * scale
            total += part
        assert (total - grad).abs().max().item() <= GREL[dname] * grad.abs().max().item()


def test_empty_and_single_row_ranges():
    from graphembed import _backend as B
    x = torch.randn(70, 6, device='cuda')
    x = x / x.norm(dim=1, keepdim=True)
    g = torch.randn(70 * 69 // 2, device='cuda')
    for form in ('valu', 'gram'):
        rc, grad = backward(form, B.SPHERE, x, g, 5, 5, True)     # no rows: a cleared gradient
        assert rc == 0 and torch.equal(grad, torch.zeros_like(grad))
        rc, grad = backward(form, B.SPHERE, x, g, 69, 70, True)   # the last row owns no pair
        assert rc == 0 and torch.equal(grad, torch.zeros_like(grad))
        rc, one = backward(form, B.SPHERE, x, g, 68, 69, True)    # one pair: (68, 69)
        assert rc == 0 and int((one.abs().sum(1) > 0).sum()) == 2


@pytest.mark.parametrize('env,select', [
    ({'MM_VEC_BWD': 'gram'}, 'test_pdist_vs_oracle_seeded or test_row_sharding or test_pdist_vs_reference_golden'),
    ({'MM_VEC_BWD': 'sym'}, 'test_pdist_vs_oracle_seeded or test_row_sharding'),
    ({'MM_VEC_LOSS_GRAM': '1'}, 'test_fused_loss_equals_unfused_path or test_tree40_training_trace_fused'),
    ({'MM_VEC_LOSS_VALU': '1'}, 'test_fused_loss_equals_unfused_path'),
    ({'MM_VEC_BWD_ORDERED': '1'}, 'test_pdist_vs_oracle_seeded or test_fused_loss_equals_unfused_path'),
])
def test_suite_with_form_forced(env, select):
    """The forms are chosen once per process (environment): the vector tests again, in a child, with the other one."""
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_vec_gpu.py'), '-x', '-q', '-m', 'gpu',
                        '-k', select, '-p', 'no:cacheprovider'], env=e, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout


# ---- the two hosts of the same pdist calls: C++ autograd nodes (default) and the torch.autograd.Function classes -----------
@pytest.mark.parametrize('case', ['spd3', 'spd4_f64', 'lorentz11', 'sphere24', 'euclidean10_plain'])
def test_cpp_autograd_nodes_equal_the_python_functions(case):
    """csrc_torch/mm_autograd.cpp issues exactly the calls of graphembed.manifolds' autograd classes: same distances bit for
    bit, same gradients up to the order of the backward's atomics; empty row ranges and shards included."""
    from graphembed import _backend as B
    from graphembed import manifolds as M
    from graphembed.manifolds.spd import _SpdPdist
    from graphembed.manifolds.vector import _VecPdist
    assert B.autograd_ext() is not None, 'lib/_mm_autograd.so is not built (python __graft_entry__.py)'
    dt = torch.float64 if case.endswith('f64') else torch.float32
    man = {'spd3': lambda: M.SymmetricPositiveDefinite(3), 'spd4_f64': lambda: M.SymmetricPositiveDefinite(4),
           'lorentz11': lambda: M.Lorentz(11), 'sphere24': lambda: M.Sphere(24), 'euclidean10_plain': lambda: M.Euclidean(10)}[case]()
    squared = not case.endswith('plain')
    torch.manual_seed(2)
    n = 301
    x = man.rand(n, out=torch.empty(0, dtype=dt, device='cuda'), ir=0.3)
    for rows in (None, (0, 0), (17, 140), (n - 1, n)):
        rb, re = rows or (0, n)
        npairs = B.pair_offset(n, re) - B.pair_offset(n, rb)
        g = torch.randn(npairs, dtype=dt, device='cuda')
        xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
        da = man.pdist(xa, squared=squared, rows=rows)
        if isinstance(man, M.SymmetricPositiveDefinite):
            db = _SpdPdist.apply(xb, man.n, squared, man.wmin, man.wmax, rb, re, man.check_pd)
        else:
            db = _VecPdist.apply(xb, man._kind, man._m, squared, rb, re, man.use_gram)
        assert da.shape == db.shape == (npairs, ) and torch.equal(da, db)
        ga, = torch.autograd.grad(da, xa, g)
        gb, = torch.autograd.grad(db, xb, g)
        assert ga.shape == gb.shape == x.shape
        scale = max(gb.abs().max().item(), 1e-30)
        assert (ga - gb).abs().max().item() <= (1e-5 if dt == torch.float32 else 1e-12) * scale
    # a non-positive-definite input surfaces as torch's LinAlgError on either host (spd.py:55-61: torch.cholesky raises)
    if isinstance(man, M.SymmetricPositiveDefinite):
        bad = x.clone()
        bad[3] = -bad[3]
        checked = M.SymmetricPositiveDefinite(man.n, check_pd=True)   # (off by default: the check reads a status word back)
        with pytest.raises(torch.linalg.LinAlgError):
            checked.pdist(bad, squared=True)
        from graphembed.manifolds.spd import _SpdPdist as py_host
        with pytest.raises(torch.linalg.LinAlgError):
            py_host.apply(bad, man.n, True, man.wmin, man.wmax, 0, n, True)


def test_suite_with_python_autograd_functions():
    """MM_PY_AUTOGRAD=1: the pdist tests of the SPD and vector suites again, in a child, through the Python classes."""
    e = dict(os.environ, MM_PY_AUTOGRAD='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_vec_gpu.py'),
                        os.path.join(ROOT, 'tests', 'test_spd_gpu.py'), '-x', '-q', '-m', 'gpu', '-k',
                        'pdist or golden or row_sharding or seamless', '-p', 'no:cacheprovider'],
                       env=e, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout


@pytest.mark.parametrize('env', [{'MM_PRODUCT_SYM': '1'}, {'MM_PRODUCT_ORDERED': '1'},
                                 {'MM_PRODUCT_ORDERED': '1', 'MM_PRODUCT_RT_KINDS': '1'}])
def test_product_suites_with_pair_kernel_forced(env):
    """(Third environment: the ordered kernel with the factor kinds read at run time instead of the instantiations that
    have them as constants — products of one or two narrow vector factors take those by default.)
    The mixed-manifold pair kernel has two forms — every ordered pair (csrc/product_pairs.hip: small n) and every
    unordered pair once (csrc/product_sym.hip: fp32 n >= 1536, fp64 n >= 640) — chosen by size: the product tests of the
    suite (golden product distances and gradients, the reference's training traces, config 4 at n = 1025 against the C
    oracle, the one-call step, a slice of the randomised campaign) again in a child with each form forced at every size."""
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_configs_gpu.py'),
                        os.path.join(ROOT, 'tests', 'test_fused_step_gpu.py'), os.path.join(ROOT, 'tests', 'test_vec_gpu.py'),
                        '-x', '-q', '-m', 'gpu', '-k', 'product or tree40 or config4', '-p', 'no:cacheprovider'],
                       env=e, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_product.py'), '60', '20281'], env=e, capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
