"""INTEGRATION.md option A, executed: this package AHEAD of the maintainer's reference checkout on PYTHONPATH.

Runs only where the reference is present (the build container; `/root/reference` does not exist on the GPU box) and in
fresh interpreters, so the `graphembed` of this test process is not touched.  What is checked is run.py's own import
block (run.py:15-18), the engines it selects (run.py:76-81), what `graphembed/__init__.py:1-9` of the reference imports,
and the dotted names `parse_config` resolves for example_config.yaml (run.py:136-206) — the hot-path names must come from
THIS package, the control plane from the checkout.  Nothing of the reference is copied: the checkout is only on the path.

tensorboard and ruamel.yaml are not in this image: `torch.utils.tensorboard` is stubbed (as SURVEY.md's Appendix B does)
and the config is read with PyYAML — the name resolution itself is run.py's `importlib.import_module` + `getattr`.
"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'matrix-manifolds_amd')
REF = '/root/reference/graphembed'

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, 'graphembed', '__init__.py')),
                                reason='the reference checkout is only present in the build container')

PRELUDE = textwrap.dedent('''
    import os, sys, types
    tb = types.ModuleType('torch.utils.tensorboard')       # (not installed here; train.py:10 imports it)
    class SummaryWriter:
        def __init__(self, *a, **k): pass
        def __getattr__(self, name): return lambda *a, **k: None
    tb.SummaryWriter = SummaryWriter
    sys.modules['torch.utils.tensorboard'] = tb
    import matplotlib
    matplotlib.use('Agg')
    OURS = os.path.realpath(os.environ['MM_PKG'])
    REF = os.path.realpath(os.environ['MM_REF'])
    def origin(obj):
        mod = sys.modules[obj.__module__] if hasattr(obj, '__module__') and not isinstance(obj, types.ModuleType) else obj
        f = os.path.realpath(getattr(mod, '__file__', None) or list(mod.__path__)[0])   # (graphembed.linalg has no __init__.py)
        return 'ours' if f.startswith(OURS + os.sep) else ('reference' if f.startswith(REF + os.sep) else f)
''')


def run(body, ours_first=True):
    env = dict(os.environ, MM_PKG=PKG, MM_REF=REF, PYTHONDONTWRITEBYTECODE='1')
    env['PYTHONPATH'] = os.pathsep.join([PKG, REF] if ours_first else [REF, PKG])
    r = subprocess.run([sys.executable, '-c', PRELUDE + textwrap.dedent(body)], env=env, capture_output=True, text=True,
                       timeout=300, cwd='/tmp')
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_run_py_import_block_resolves_across_both_packages():
    out = run('''
        from graphembed.data import load_graph_pdists, GraphDataset                      # run.py:15
        from graphembed.products.embedding import Embedding as ProductManifoldEmbedding   # run.py:16
        from graphembed.pyx import FastPrecision                                          # run.py:17
        from graphembed.utils import check_mkdir, nnm1d2_to_n, Timer                      # run.py:18
        from graphembed.products import TrainingEngine as PE                              # run.py:77
        from graphembed.train_da import TrainingEngine as DA                              # run.py:79
        from graphembed.train import TrainingEngine                                       # run.py:81
        import graphembed
        from graphembed import linalg, manifolds, optim, modules, objectives, train, data  # reference __init__.py:1-9
        import graphembed.monitor, graphembed.inference
        for name, obj in [('GraphDataset', GraphDataset), ('load_graph_pdists', load_graph_pdists),
                          ('FastPrecision', FastPrecision), ('check_mkdir', check_mkdir),
                          ('manifolds', manifolds), ('optim', optim), ('modules', modules), ('objectives', objectives),
                          ('SPD', manifolds.SymmetricPositiveDefinite), ('RSGD', optim.RiemannianSGD),
                          ('ManifoldEmbedding', modules.ManifoldEmbedding), ('BatchedObjective', modules.BatchedObjective)]:
            assert origin(obj) == 'ours', (name, origin(obj))
        for name, obj in [('ProductManifoldEmbedding', ProductManifoldEmbedding), ('TrainingEngine', TrainingEngine),
                          ('products.TrainingEngine', PE), ('train_da.TrainingEngine', DA),
                          ('monitor', graphembed.monitor), ('inference', graphembed.inference),
                          ('Universal', manifolds.Universal), ('OrthogonalGroup', manifolds.OrthogonalGroup),
                          ('EmbeddingBase', modules.EmbeddingBase)]:
            assert origin(obj) == 'reference', (name, origin(obj))
        # graphembed.linalg has no __init__.py on either side: a namespace package over both directories — `fast` (the closed
        # forms on the path, csrc/fast.hip) from this package, `torch_batch` (CPU-offloaded LAPACK wrappers) from the checkout
        from graphembed.linalg import fast, torch_batch as tb
        assert origin(fast) == 'ours' and origin(tb) == 'reference', (origin(fast), origin(tb))
        assert [os.path.realpath(p).startswith(OURS) for p in linalg.__path__] == [True, False], list(linalg.__path__)
        from graphembed.utils import PLT_MUTEX, latest_path_by_basename_numeric_order     # train.py:15-16 (names only the checkout has)
        assert origin(graphembed._overlay._counterpart('utils')) == 'reference'
        # the engine's own base class check: the checkout's TrainingEngine drives THIS package's BatchedObjective
        import inspect
        assert 'BatchedObjective' in inspect.getsource(train)
        assert train.BatchedObjective is modules.BatchedObjective
        # the reference's Universal manifold subclasses THIS package's Manifold (one class hierarchy, not two)
        assert issubclass(manifolds.Universal, manifolds.Manifold) and origin(manifolds.Manifold) == 'ours'
        print('ok')
    ''')
    assert out.strip().endswith('ok')


def test_attribute_misses_stay_attribute_errors():
    """An attribute MISS on the overlaid package must not import (and fail in) checkout modules that merely mention the
    name: `hasattr(graphembed, 'torch')` used to raise ModuleNotFoundError('tensorboard') out of the reference's train.py
    (advisor, round 4).  Run WITHOUT the tensorboard stub of the other tests."""
    env = dict(os.environ, MM_PKG=PKG, MM_REF=REF, PYTHONDONTWRITEBYTECODE='1', PYTHONPATH=os.pathsep.join([PKG, REF]))
    code = textwrap.dedent('''
        import sys
        import graphembed, graphembed._overlay as o
        assert len(o.later_packages()) == 1
        assert 'torch.utils.tensorboard' not in sys.modules
        assert not hasattr(graphembed, 'torch') and not hasattr(graphembed, 'np')      # (mentioned all over the checkout)
        for name in ('definitely_not_there', 'pytest_plugins', '_pytestfixturefunction', 'SummaryWriter'):
            assert not hasattr(graphembed, name), name
            assert getattr(graphembed.manifolds, name, None) is None, name
            assert getattr(graphembed.modules, name, 7) == 7, name
        assert not any(m.startswith('graphembed.train') for m in sys.modules), [m for m in sys.modules if 'train' in m]
        # a module asked for BY NAME whose own imports are missing here: an AttributeError that names the cause
        try:
            graphembed.train
        except AttributeError as e:
            assert 'tensorboard' in str(e), e
        else:
            raise AssertionError('graphembed.train imported without tensorboard?')
        # names the checkout really binds still resolve
        from graphembed.manifolds import Universal
        assert o.later_packages() is not o.later_packages() and o._later_cache      # cached scan, fresh list
        print('ok')
    ''')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300, cwd='/tmp')
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout + r.stderr


def test_linalg_fast_hands_cpu_tensors_to_the_checkout():
    """graphembed.linalg.fast of this package shadows the checkout's: its CPU callers (monitor.py, tests/test_linalg.py) get
    the checkout's own functions for CPU tensors (advisor, round 4); GPU tensors stay on the kernels."""
    out = run('''
        import torch
        from graphembed.linalg import fast
        assert origin(fast) == 'ours'
        torch.manual_seed(0)
        a = torch.randn(7, 3, 3, dtype=torch.float64)
        x = a @ a.transpose(1, 2) + torch.eye(3, dtype=torch.float64)
        w = fast.symeig3x3(x)
        assert torch.allclose(w, torch.linalg.eigvalsh(x), atol=1e-6), (w, torch.linalg.eigvalsh(x))
        x2 = x[:, :2, :2].clone().requires_grad_()
        l = fast.cholesky2x2(x2)
        assert torch.allclose(l @ l.transpose(1, 2), x2, atol=1e-6)
        g, = torch.autograd.grad(l.sum(), x2, create_graph=True)       # the checkout's pure-torch form: twice differentiable
        assert g.requires_grad
        li, lc = fast.invcholesky2x2(x2.detach(), ret_chol=True)
        assert torch.allclose(li @ lc, torch.eye(2, dtype=torch.float64).expand(7, 2, 2), atol=1e-6)
        assert origin(fast._reference_fast()) == 'reference'
        print('ok')
    ''')
    assert out.strip().endswith('ok')


def test_example_config_names_resolve_like_parse_config():
    out = run('''
        import importlib, yaml
        cfg = yaml.safe_load(open(os.path.join(REF, 'example_config.yaml')))
        names = []
        def walk(node):
            if isinstance(node, dict):
                for k, v in node.items():
                    if k in ('object', 'closure'):
                        names.append(v['name'])
                    walk(v)
            elif isinstance(node, list):
                for v in node:
                    walk(v)
        walk(cfg)
        assert names, cfg
        got = {}
        for dotted in names:                                   # run.py:188-192
            parts = dotted.split('.')
            module = importlib.import_module('.'.join(parts[:-1]))
            got[dotted] = origin(getattr(module, parts[-1]))
        print(sorted(got.items()))
        assert got['graphembed.modules.ManifoldEmbedding'] == 'ours'
        assert got['graphembed.manifolds.SymmetricPositiveDefinite'] == 'ours'
        assert got['graphembed.optim.RiemannianAdam'] == 'ours'
        assert got['graphembed.objectives.KLDiveregenceLoss'] == 'reference'   # (not on the hot path: SURVEY.md section 2)
        print('ok')
    ''')
    assert out.strip().endswith('ok')


def test_without_a_checkout_behind_it_the_package_is_unchanged():
    env = dict(os.environ, PYTHONPATH=PKG, PYTHONDONTWRITEBYTECODE='1')
    code = textwrap.dedent('''
        import graphembed, graphembed._overlay as o
        assert o.later_packages() == [] and len(graphembed.__path__) == 1
        try:
            import graphembed.train
        except ModuleNotFoundError:
            print('ok')
    ''')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300, cwd='/tmp')
    assert r.returncode == 0 and r.stdout.strip() == 'ok', r.stdout + r.stderr
