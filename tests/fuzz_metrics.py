#!/usr/bin/env python3
"""Randomised cross-check of the GPU evaluator (graphembed.pyx.FastPrecision: MAP and the layer-wise F1 curves)
against the numpy restatements in oracle/ref_port.py on random graphs (trees, sparse / dense Erdos-Renyi, paths,
stars), random embedding distances with heavy ties, degree filters, both aggregations.
Not collected by pytest (run by hand on a GPU box): python tests/fuzz_metrics.py [cases] [seed]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import networkx as nx  # noqa: E402
import torch  # noqa: E402
from scipy.sparse.csgraph import shortest_path  # noqa: E402
from scipy.spatial.distance import squareform  # noqa: E402
from graphembed.pyx import FastPrecision  # noqa: E402
from oracle import ref_port as rp  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    for c in range(cases):
        kind = rng.choice(['tree', 'er_sparse', 'er_dense', 'path', 'star', 'cycle'])
        n = rng.choice([2, 3, 5, 17, 64, 65, 255, 256, 257, rng.randint(2, 400)])
        seed = rng.randint(0, 10**6)
        if kind == 'tree':
            g = nx.random_labeled_tree(n, seed=seed) if hasattr(nx, 'random_labeled_tree') else nx.random_tree(n, seed=seed)
        elif kind == 'path':
            g = nx.path_graph(n)
        elif kind == 'star':
            g = nx.star_graph(n - 1)
        elif kind == 'cycle':
            g = nx.cycle_graph(max(n, 3))
        else:
            g = nx.erdos_renyi_graph(n, (2.5 / max(n, 2)) if kind == 'er_sparse' else 0.4, seed=seed)
            g = g.subgraph(max(nx.connected_components(g), key=len)).copy()
        g = nx.convert_node_labels_to_integers(g)
        m = g.number_of_nodes()
        if m < 2:
            continue
        hops = shortest_path(nx.to_scipy_sparse_array(g, nodelist=range(m)), unweighted=True).astype(np.int64)
        nrng = np.random.default_rng(seed)
        mode = rng.choice(['random', 'ties', 'quantised', 'perfect'])
        P = m * (m - 1) // 2
        if mode == 'perfect':
            pd = squareform(hops.astype(np.float64), checks=False)
        else:
            pd = nrng.random(P)
            if mode == 'ties':
                pd[::3] = pd[0]
            if mode == 'quantised':
                pd = np.round(pd * 4) / 4 + 0.25
        dtype = rng.choice([np.float32, np.float64])
        pd = pd.astype(dtype)
        dense = squareform(pd.astype(np.float64))
        nb = [set(g.neighbors(u)) for u in range(m)]
        deg = np.array([g.degree(u) for u in range(m)])
        fp = FastPrecision(g)
        t = torch.from_numpy(pd)
        got = fp.mean_average_precision(t)
        ref = rp.mean_average_precision(dense, nb)
        if abs(got - ref) > 1e-6:
            print(f'FAIL case {c}: MAP {kind} n={m} {mode} {dtype.__name__}: {got} vs {ref}')
            sys.exit(1)
        kw = {}
        if rng.random() < 0.4:
            lo = rng.randint(0, 3)
            kw = dict(min_degree=lo, max_degree=lo + rng.randint(0, 6))
            if not ((deg >= kw['min_degree']) & (deg <= kw['max_degree'])).any():
                kw = {}
        for avg in (False, True):
            fn = fp.layer_mean_average_f1_scores if avg else fp.layer_mean_f1_scores
            k2 = {} if avg else kw   # the per-tree average has no degree filter (pyx/precision.pyx)
            means, var = fn(t, **k2)
            rmeans, rvar = rp.layer_f1_scores(dense, hops, per_tree_average=avg, degrees=deg, **k2)
            if not (np.allclose(means, rmeans, rtol=1e-9, atol=1e-11, equal_nan=True)
                    and np.allclose(var, rvar, rtol=1e-7, atol=1e-11, equal_nan=True)):
                print('got ', np.asarray(means)[:12], np.asarray(var)[:6])
                print('want', np.asarray(rmeans)[:12], np.asarray(rvar)[:6])
                print(f'FAIL case {c}: F1 avg={avg} {kind} n={m} {mode} {dtype.__name__} kw={kw}: '
                      f'{np.abs(np.asarray(means) - rmeans).max():.2e} {np.abs(np.asarray(var) - rvar).max():.2e}')
                sys.exit(1)
    print(f'{cases} cases ok')


if __name__ == '__main__':
    main()
