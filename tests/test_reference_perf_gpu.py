"""The reference's performance tests (graphembed/tests/test_perf.py) on this package's classes, with the reference's own
sizes and its GPU time limits: a million 2x2 / 3x3 eigenvalue problems, pdist against dist on the gathered pairs, the native
MAP evaluator against the pure-Python one."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def timeit(fn, number):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(number):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


@pytest.mark.parametrize('d', [2, 3])
def test_symeig_of_a_million_matrices(d):
    """test_perf.py:29-40: 10 calls on 10^6 random symmetric d x d matrices within the reference's GPU limit of 0.1 s."""
    from graphembed.manifolds import SymmetricPositiveDefinite as SPD
    torch.manual_seed(0)
    x = torch.rand(1000000, d, d, device='cuda')
    x = 0.5 * (x + x.transpose(1, 2))
    spd = SPD(d)
    assert timeit(lambda: spd.symeig(x), number=10) < 0.1


@pytest.mark.parametrize('d,n', [(2, 5000), (3, 1000)])
def test_spd_pdist_faster_than_dist_on_gathered_pairs(d, n):
    """test_perf.py:84-106 (its GPU sizes)."""
    from graphembed.manifolds import SymmetricPositiveDefinite as SPD
    torch.manual_seed(0)
    a = torch.rand(n, d, d, device='cuda')
    x = a @ a.transpose(1, 2) + torch.eye(d, device='cuda')
    m = torch.triu_indices(n, n, 1, device='cuda')
    spd = SPD(d)
    t_pdist = timeit(lambda: spd.pdist(x), number=10)
    t_dist = timeit(lambda: spd.dist(x[m[0]], x[m[1]]), number=10)
    assert t_pdist < t_dist
    np.testing.assert_allclose(spd.pdist(x).cpu().numpy(), spd.dist(x[m[0]], x[m[1]]).cpu().numpy(), rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize('n', [500, 2000])
def test_native_map_faster_than_python(n):
    """test_perf.py:109-127: FastPrecision.mean_average_precision against the pure-Python MAP on an Erdos-Renyi graph."""
    import networkx as nx
    from scipy.spatial.distance import squareform
    from graphembed.pyx import FastPrecision
    from oracle import ref_port as rp
    g = nx.erdos_renyi_graph(n, 0.1, seed=1)
    g = nx.convert_node_labels_to_integers(g.subgraph(max(nx.connected_components(g), key=len)).copy())
    n = g.number_of_nodes()
    pd = np.random.default_rng(0).random(n * (n - 1) // 2).astype(np.float32)
    dense = squareform(pd)
    nb = [set(g.neighbors(u)) for u in range(n)]
    t0 = time.perf_counter()
    ref = rp.mean_average_precision(dense, nb)          # the oracle's restatement of py_mean_average_precision
    t_py = time.perf_counter() - t0
    fp = FastPrecision(g)
    pdt = torch.from_numpy(pd)
    t_fp = timeit(lambda: fp.mean_average_precision(pdt), number=1)
    assert t_fp < t_py
    assert abs(fp.mean_average_precision(pdt) - ref) <= 1e-6
