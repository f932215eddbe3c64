"""Round 3, GPU: the symmetric-tile Gram backward in both super-tile shapes (a per-process choice, so each runs in its
own interpreter)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('nw', ['3', '4', 'auto'])
def test_gram_backward_super_tile_shapes(nw):
    env = dict(os.environ)
    env.pop('MM_GRAM_BWD_NW', None)
    if nw != 'auto':
        env['MM_GRAM_BWD_NW'] = nw
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'gram_shapes_check.py')], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert 'gram shapes ok' in r.stdout
