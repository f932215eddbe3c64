import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'matrix-manifolds_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run via gpurun)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


_cache = {}


def load_golden(name):
    if name not in _cache:
        with np.load(os.path.join(GOLDEN, name + '.npz')) as z:
            _cache[name] = {k: z[k] for k in z.files}
    return _cache[name]


@pytest.fixture(scope='session')
def golden():
    return load_golden


def sym(a):
    return 0.5 * (a + np.swapaxes(a, -1, -2))
