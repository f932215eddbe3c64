"""Metrics: the oracle's MAP restatement and this package's metric functions against
values recorded from the reference's graphembed/metrics.py; the GPU average-precision kernel against
both (the reference's own tests: tests/test_metrics.py:15-24)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

CASES = [f'er{n}_{p}_{s}' for n, p in ((50, 0.1), (100, 0.05), (100, 0.5), (300, 0.02)) for s in (0, 1)]


def _graph(G, tag):
    import networkx as nx
    g = nx.Graph()
    g.add_nodes_from(range(int(G[f'{tag}/n'])))
    g.add_edges_from(np.asarray(G[f'{tag}/edges']).tolist())
    return g


@pytest.mark.parametrize('tag', CASES)
def test_oracle_map_vs_reference(tag):
    from scipy.spatial.distance import squareform
    from oracle import ref_port as rp
    G = load_golden('metrics')
    g = _graph(G, tag)
    nb = [set(g.neighbors(u)) for u in range(g.number_of_nodes())]
    got = rp.mean_average_precision(squareform(np.asarray(G[f'{tag}/pdists'], dtype=np.float64)), nb)
    assert abs(got - float(G[f'{tag}/map'])) <= 1e-12


@pytest.mark.parametrize('tag', CASES[:4])
def test_scalar_metrics_vs_reference(tag):
    from graphembed import metrics as M
    G = load_golden('metrics')
    md, gd = torch.from_numpy(np.array(G[f'{tag}/md'])), torch.from_numpy(np.array(G[f'{tag}/gd']))
    assert abs(M.average_distortion(md, gd).item() - float(G[f'{tag}/distortion'])) <= 1e-12
    assert abs(M.pearsonr(md, gd).item() - float(G[f'{tag}/pearsonr'])) <= 1e-12
    assert abs(M.average_pearsonr(md, gd).item() - float(G[f'{tag}/avg_pearsonr'])) <= 1e-10
    vs = np.array(G['auc/vs'])
    np.testing.assert_allclose(M.area_under_curve(vs), G['auc/full'], rtol=1e-12)
    np.testing.assert_allclose(M.area_under_curve(vs, 4), G['auc/step4'], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', CASES)
def test_gpu_map_vs_reference_golden(tag):
    from graphembed.metrics import mean_average_precision
    from graphembed.pyx import FastPrecision
    G = load_golden('metrics')
    g = _graph(G, tag)
    pd = torch.from_numpy(np.array(G[f'{tag}/pdists']))
    ref = float(G[f'{tag}/map'])
    assert abs(FastPrecision(g).mean_average_precision(pd) - ref) <= 1e-6
    assert abs(mean_average_precision(pd.double().cuda(), g) - ref) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize('n,p', [(1000, 0.01), (1000, 0.5), (2500, 0.004)])
def test_gpu_map_vs_oracle_seeded(n, p):
    """The reference's test_cython_map sizes: random distances on Erdos-Renyi graphs; ties (duplicated
    distances) resolve by node index like a stable argsort; the perfect embedding scores 1."""
    import networkx as nx
    from scipy.spatial.distance import squareform
    from graphembed.data.graph import compute_graph_pdists
    from graphembed.pyx import FastPrecision
    from oracle import ref_port as rp
    g = nx.erdos_renyi_graph(n, p, seed=3)
    g = nx.convert_node_labels_to_integers(g.subgraph(max(nx.connected_components(g), key=len)).copy())
    m = g.number_of_nodes()
    rng = np.random.default_rng(5)
    pd = rng.random(m * (m - 1) // 2).astype(np.float32)
    pd[::7] = pd[3]  # ties
    nb = [set(g.neighbors(u)) for u in range(m)]
    fp = FastPrecision(g)
    ref = rp.mean_average_precision(squareform(pd.astype(np.float64)), nb)
    assert abs(fp.mean_average_precision(torch.from_numpy(pd)) - ref) <= 1e-6
    assert abs(fp.mean_average_precision(torch.from_numpy(compute_graph_pdists(g)).float()) - 1.0) <= 1e-6


def _er(n, p, seed):
    import networkx as nx
    g = nx.erdos_renyi_graph(n, p, seed=seed)
    return nx.convert_node_labels_to_integers(g.subgraph(max(nx.connected_components(g), key=len)).copy())


def _hops(g):
    import networkx as nx
    from scipy.sparse.csgraph import shortest_path
    return shortest_path(nx.to_scipy_sparse_array(g, nodelist=range(len(g))), unweighted=True).astype(np.int64)


@pytest.mark.parametrize('n,p', [(50, 0.1), (120, 0.04)])
def test_oracle_f1_known_answer(n, p):
    """The reference's own known-answer test (tests/test_metrics.py:27-35): with the graph distances as the
    embedding distances every layer scores F1 = 1 — whatever the order inside the tied layers."""
    from oracle import ref_port as rp
    g = _er(n, p, 1)
    hops = _hops(g)
    for avg in (False, True):
        means, var = rp.layer_f1_scores(hops.astype(np.float64), hops, per_tree_average=avg)
        np.testing.assert_allclose(means, np.ones_like(means), atol=1e-12)
        np.testing.assert_allclose(var, np.zeros_like(var), atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('n,p', [(50, 0.1), (500, 0.01), (500, 0.1)])
def test_gpu_f1_trivial(n, p):
    """test_f1_trivial of the reference (tests/test_metrics.py:27-35) on the GPU evaluator."""
    from graphembed.data.graph import compute_graph_pdists
    from graphembed.metrics import area_under_curve
    from graphembed.pyx import FastPrecision
    g = _er(n, p, 0)
    fp = FastPrecision(g)
    pd = torch.from_numpy(compute_graph_pdists(g)).float()
    means, _ = fp.layer_mean_f1_scores(pd)
    np.testing.assert_allclose(means, np.ones(len(means)), atol=1e-6)
    np.testing.assert_allclose(area_under_curve(means), 1, atol=1e-6)
    means, _ = fp.layer_mean_average_f1_scores(pd)
    np.testing.assert_allclose(means, np.ones(len(means)), atol=1e-6)
    assert fp.nodes_per_layer()[0] == g.number_of_nodes() and sum(fp.nodes_per_layer()) == g.number_of_nodes() ** 2


@pytest.mark.gpu
@pytest.mark.parametrize('n,p', [(80, 0.08), (300, 0.02)])
def test_gpu_f1_vs_oracle(n, p):
    """Random embedding distances (with ties): GPU layer F1 curves vs the numpy restatement of
    precision.cpp:300-429, both aggregations, the degree filter and two concatenated distance sets."""
    from scipy.spatial.distance import squareform
    from graphembed.pyx import FastPrecision
    from oracle import ref_port as rp
    g = _er(n, p, 2)
    m = g.number_of_nodes()
    hops = _hops(g)
    rng = np.random.default_rng(4)
    pd = rng.random(m * (m - 1) // 2)
    pd[::5] = pd[2]
    dense = squareform(pd)
    deg = np.array([g.degree(u) for u in range(m)])
    fp = FastPrecision(g)
    for kw, okw in ((dict(), dict(degrees=deg)), (dict(min_degree=2, max_degree=6), dict(degrees=deg, min_degree=2, max_degree=6))):
        means, var = fp.layer_mean_f1_scores(torch.from_numpy(pd), **kw)
        rmeans, rvar = rp.layer_f1_scores(dense, hops, **okw)
        np.testing.assert_allclose(means, rmeans, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(var, rvar, rtol=1e-8, atol=1e-12)
    means, var = fp.layer_mean_average_f1_scores(torch.from_numpy(pd))
    rmeans, rvar = rp.layer_f1_scores(dense, hops, per_tree_average=True)
    np.testing.assert_allclose(means, rmeans, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(var, rvar, rtol=1e-8, atol=1e-12)
    pd2 = rng.random(pd.shape)
    means2, _ = fp.layer_mean_f1_scores(torch.from_numpy(np.concatenate([pd, pd2])), num_pdists_sets=2)
    a, _ = rp.layer_f1_scores(dense, hops)
    b, _ = rp.layer_f1_scores(squareform(pd2), hops)
    np.testing.assert_allclose(means2, 0.5 * (a + b), rtol=1e-10)


@pytest.mark.gpu
def test_gpu_f1_long_paths():
    """Graphs with more than 256 layers (a path, a ring: diameter >= 256) take the large-capacity instantiation
    of the layer-F1 kernel — found by tests/fuzz_metrics.py as an `unsupported` error."""
    import networkx as nx
    from scipy.spatial.distance import squareform
    from graphembed.pyx import FastPrecision
    from oracle import ref_port as rp
    for g in (nx.path_graph(300), nx.cycle_graph(700)):
        m = g.number_of_nodes()
        hops = _hops(g)
        rng = np.random.default_rng(1)
        pd = rng.random(m * (m - 1) // 2)
        fp = FastPrecision(g)
        means, var = fp.layer_mean_f1_scores(torch.from_numpy(pd))
        rmeans, rvar = rp.layer_f1_scores(squareform(pd), hops, degrees=np.array([g.degree(u) for u in range(m)]))
        np.testing.assert_allclose(means, rmeans, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(var, rvar, rtol=1e-8, atol=1e-12)


# ------------------------------------------------------------------ weighted graphs (precision.cpp:76-92, 150-166)
def _weighted_er(n, p, seed, wmax=4):
    g = _er(n, p, seed)
    rng = np.random.default_rng(seed + 100)
    for u, v in g.edges():
        g[u][v]['weight'] = int(rng.integers(1, wmax + 1))
    return g


def _weighted_layers(g):
    """Layers of every shortest-path tree of a weighted graph, restated from precision.cpp:76-92 (Dijkstra with a heap)
    and 150-166 (a new layer wherever the sorted distances increase) — plain Python, independent of the package."""
    import heapq
    n = len(g)
    dist = np.zeros((n, n), dtype=np.int64)
    layers = np.zeros((n, n), dtype=np.int64)
    for u in range(n):
        d = [np.iinfo(np.int64).max] * n
        d[u] = 0
        q = [(0, u)]
        while q:
            du, node = heapq.heappop(q)
            if du != d[node]:
                continue
            for nb, attr in g[node].items():
                nd = du + int(attr['weight'])
                if nd < d[nb]:
                    d[nb] = nd
                    heapq.heappush(q, (nd, nb))
        dist[u] = d
        order = sorted(range(n), key=lambda v: d[v])
        layer = 0
        for i in range(1, n):
            if d[order[i]] > d[order[i - 1]]:
                layer += 1
            layers[u, order[i]] = layer
    return dist, layers


def test_weighted_layers_host_side():
    """graphembed.pyx.tree_layers (dense ranks of the scipy Dijkstra distances) == the reference's construction."""
    import networkx as nx
    from scipy.sparse.csgraph import shortest_path
    from graphembed.pyx import tree_layers
    g = _weighted_er(60, 0.08, 3)
    dist, layers = _weighted_layers(g)
    sp = shortest_path(nx.to_scipy_sparse_array(g, nodelist=range(len(g)), weight='weight'), method='D')
    np.testing.assert_array_equal(sp.astype(np.int64), dist)
    np.testing.assert_array_equal(tree_layers(sp), layers)
    # unweighted: the layers are the hop distances
    h = _hops(_er(60, 0.08, 3))
    np.testing.assert_array_equal(tree_layers(h), h)


@pytest.mark.gpu
@pytest.mark.parametrize('n,p', [(70, 0.08), (250, 0.03)])
def test_gpu_f1_weighted_graph(n, p):
    """Weighted graphs: the perfect embedding (embedding distance = weighted graph distance) scores F1 = 1 on every
    layer (the reference's known answer, tests/test_metrics.py:27-35, carried over), random distances agree with the
    numpy restatement on the weighted layers, and MAP only looks at the adjacency."""
    from scipy.spatial.distance import squareform
    from graphembed.pyx import FastPrecision
    from oracle import ref_port as rp
    g = _weighted_er(n, p, 5)
    m = g.number_of_nodes()
    dist, layers = _weighted_layers(g)
    fp = FastPrecision(g)
    assert fp.weighted and fp.num_layers == layers.max() + 1
    assert fp.nodes_per_layer() == [int(c) for c in np.bincount(layers.reshape(-1))]
    perfect = torch.from_numpy(squareform(dist.astype(np.float64), checks=False))
    for means, var in (fp.layer_mean_f1_scores(perfect), fp.layer_mean_average_f1_scores(perfect)):
        np.testing.assert_allclose(means[np.isfinite(means)], 1.0, atol=1e-10)
    rng = np.random.default_rng(6)
    pd = rng.random(m * (m - 1) // 2)
    pd[::4] = pd[1]
    means, var = fp.layer_mean_f1_scores(torch.from_numpy(pd))
    rmeans, rvar = rp.layer_f1_scores(squareform(pd), layers, degrees=np.array([g.degree(u) for u in range(m)]))
    np.testing.assert_allclose(means, rmeans, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(var, rvar, rtol=1e-8, atol=1e-12)
    nb = [set(g.neighbors(u)) for u in range(m)]
    assert abs(fp.mean_average_precision(torch.from_numpy(pd)) - rp.mean_average_precision(squareform(pd), nb)) <= 1e-9


@pytest.mark.gpu
def test_gpu_row_sort_is_a_stable_argsort():
    """mm_graph_sort_rows == a stable argsort of every row (ties in node order), fp32 and fp64."""
    from graphembed import _backend as B
    lib = B.lib()
    for dt in (torch.float32, torch.float64):
        n = 777
        torch.manual_seed(1)
        d = torch.rand(n, n, dtype=dt, device='cuda')
        d[:, ::3] = d[:, :1]          # many ties per row
        d[5] = 0.25
        code = B.dtype_code(d)
        nbytes = lib.raw('mm_graph_sort_rows_ws_bytes')(code, n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
        order = torch.empty(n, n, dtype=torch.int32, device='cuda')
        lib.call('mm_graph_sort_rows', code, B.ptr(d), n, B.ptr(order), B.ptr(ws), nbytes, B.stream_of(d))
        want = torch.sort(d, dim=1, stable=True).indices.to(torch.int32)
        assert torch.equal(order, want)
