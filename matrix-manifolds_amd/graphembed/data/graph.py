"""Graph loading and all-pairs shortest paths — counterpart of
graphembed/graphembed/data/graph.py:15-87 (edge lists / .npy; APSP by scipy's BFS,
the reference's networkit is not a dependency here)."""
import gzip
import os

import numpy as np
import torch

CACHED_PDISTS_FILE = 'cached_pdists.npy'


def compute_graph_pdists(g, cache_dir=None):
    import networkx as nx
    from scipy.sparse.csgraph import shortest_path
    from scipy.spatial.distance import squareform
    a = nx.to_scipy_sparse_array(g, nodelist=range(len(g)))
    d = shortest_path(a, unweighted=True, directed=g.is_directed())
    pd = squareform(d, checks=False)
    if cache_dir:
        np.save(os.path.join(cache_dir, CACHED_PDISTS_FILE), pd)
    return pd


def load_graph_pdists(f, cache_dir=None, flip_probability=None):
    """Returns (condensed shortest-path distances as a tensor, networkx graph or None)."""
    import networkx as nx
    if flip_probability is not None:
        raise NotImplementedError('noisy-graph generation is outside the accelerated path')
    f = os.path.abspath(os.path.realpath(f))
    if f.endswith('.npy'):
        return torch.from_numpy(np.load(f)).to(torch.get_default_dtype()), None
    directed = f.endswith('.dir-edges') or f.endswith('.dir-edges.gz')
    opener = gzip.open if f.endswith('.gz') else open
    with opener(f, 'rt') as fh:
        g = nx.parse_edgelist((l for l in fh if l.strip() and not l.startswith('#')),
                              create_using=nx.DiGraph if directed else nx.Graph, data=False)
    g = nx.convert_node_labels_to_integers(g)
    assert g.is_directed() or nx.number_connected_components(g) == 1
    cache = None
    if cache_dir is not None:
        cache = os.path.join(cache_dir, os.path.basename(f))
        cached = os.path.join(cache, CACHED_PDISTS_FILE)
        if os.path.isfile(cached):
            return torch.from_numpy(np.load(cached)).to(torch.get_default_dtype()), g
        os.makedirs(cache, exist_ok=True)
    pd = compute_graph_pdists(g, cache)
    return torch.from_numpy(pd).to(torch.get_default_dtype()), g
