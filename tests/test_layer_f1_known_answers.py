"""Known answers for the layer-wise F1 curves, derived BY HAND from the definition in the reference's only native component
(graphembed/graphembed/pyx/impl/precision.cpp:321-398 `LayerF1Scores`, :401-458 the two aggregations) — that file cannot be
compiled here (boost flat_set, TBB), and the reference's own test pins only the trivial case F1 == 1
(tests/test_metrics.py:27-35).  These cases have precision < 1 and recall < 1, several layers, the per-tree aggregation and
the degree filter; both the oracle restatement (oracle/ref_port.py, CPU) and the GPU evaluator are held to them.

The definition, for a root u and the i-th node v (i = 1 .. n-1) in the order of the EMBEDDING distances from u, with
layer(v) = the hop distance u -> v:
    before    = 1 + #{nodes seen earlier whose layer <= layer(v)}                      (:352-356, the multiset insert)
    precision = before / i                                                               (:359)
    actual    = #{nodes on layers 1 .. layer(v)-1} + #{seen earlier on layer(v)} + 1     (:361-370)
    recall    = before / actual                                                          (:374)
    f1        = 2 precision recall / (precision + recall)  -> accumulated under layer(v) (:376-379)
LayerMeanF1Scores: mean and E[f^2] - mean^2 over all (u, v) of a layer, roots filtered by degree (:401-428);
LayerMeanAverageF1Scores: the same over the per-root layer means (:430-458).

Every derivation is written out below as exact fractions."""
import os
import sys
from fractions import Fraction as F

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def moments(values):
    m = sum(values, F(0)) / len(values)
    return m, sum((v * v for v in values), F(0)) / len(values) - m * m


# ---- case 1: the path 0 - 1 - 2, embedding distances d01 = 3, d02 = 1, d12 = 2 (the embedding puts 0 next to 2) ----------
# root 0 (layers: 1 -> 1, 2 -> 2), embedding order 2, 1:
#   v = 2 (layer 2): before 1, precision 1/1, actual 1 + 0 + 1 = 2, recall 1/2        -> f1 = 2/3   [layer 2]
#   v = 1 (layer 1): before 1 (node 2 is on a later layer), precision 1/2, actual 1, recall 1 -> f1 = 2/3   [layer 1]
# root 1 (both on layer 1), order 2, 0:   v = 2: 1, 1/1, actual 1 -> f1 = 1;   v = 0: before 2, 2/2, actual 0 + 1 + 1 = 2 -> f1 = 1
# root 2 (layers: 1 -> 1, 0 -> 2), order 0, 1:   v = 0 (layer 2): 1, 1, actual 2, recall 1/2 -> 2/3;   v = 1 (layer 1): 1, 1/2, 1 -> 2/3
PATH3 = dict(
    edges=[(0, 1), (1, 2)], pdists=[3.0, 1.0, 2.0],
    all_nodes=[moments([F(2, 3), F(1), F(1), F(2, 3)]), moments([F(2, 3), F(2, 3)])],          # layer 1: 5/6, 1/36; layer 2: 2/3, 0
    per_tree=[moments([F(2, 3), F(1), F(2, 3)]), moments([F(2, 3), F(2, 3)])])                  # layer 1: 7/9, 2/81
assert PATH3['all_nodes'] == [(F(5, 6), F(1, 36)), (F(2, 3), F(0))] and PATH3['per_tree'][0] == (F(7, 9), F(2, 81))

# ---- case 2: the star with centre 0 and leaves 1, 2, 3; d01 = 1, d02 = 2, d03 = 3, d12 = 1.5, d13 = 2.5, d23 = 0.5 ------------
# root 0 (all leaves on layer 1), order 1, 2, 3: before = i, precision 1, actual = seen-on-the-layer + 1 = i -> f1 = 1, 1, 1
# root 1 (0 -> layer 1; 2, 3 -> layer 2), order 0, 2, 3:
#   v = 0: 1, 1, actual 1 -> 1 [layer 1];  v = 2: before 2, 2/2, actual 1 + 0 + 1 = 2 -> 1 [layer 2];  v = 3: before 3, 3/3, actual 1 + 1 + 1 -> 1 [layer 2]
# root 2 (0 -> 1; 1, 3 -> 2), order 3, 1, 0:
#   v = 3 (layer 2): before 1, precision 1, actual 1 + 0 + 1 = 2, recall 1/2              -> 2/3
#   v = 1 (layer 2): before 2, precision 2/2, actual 1 + 1 + 1 = 3, recall 2/3            -> 2 (2/3) / (5/3) = 4/5
#   v = 0 (layer 1): before 1 (both seen nodes are on layer 2), precision 1/3, actual 1, recall 1 -> 2 (1/3) / (4/3) = 1/2
# root 3 (0 -> 1; 1, 2 -> 2), order 2, 1, 0: the same pattern -> 2/3, 4/5 [layer 2], 1/2 [layer 1]
STAR4 = dict(
    edges=[(0, 1), (0, 2), (0, 3)], pdists=[1.0, 2.0, 3.0, 1.5, 2.5, 0.5],
    all_nodes=[moments([F(1), F(1), F(1), F(1), F(1, 2), F(1, 2)]), moments([F(1), F(1), F(2, 3), F(4, 5), F(2, 3), F(4, 5)])],
    per_tree=[moments([F(1), F(1), F(1, 2), F(1, 2)]), moments([F(1), F(11, 15), F(11, 15)])],   # (the centre has no layer 2)
    leaves_only=[moments([F(1), F(1, 2), F(1, 2)]), moments([F(1), F(1), F(2, 3), F(4, 5), F(2, 3), F(4, 5)])])   # max_degree = 1
assert STAR4['all_nodes'] == [(F(5, 6), F(1, 18)), (F(37, 45), F(38, 2025))]
assert STAR4['per_tree'] == [(F(3, 4), F(1, 16)), (F(37, 45), F(32, 2025))] and STAR4['leaves_only'][0] == (F(2, 3), F(1, 18))

# ---- case 3: the ring 0 - 1 - 2 - 3 - 0; d01 = 1, d02 = 1.2, d03 = 3, d12 = 2, d13 = 0.7, d23 = 1.5 --------------------------
# root 0 (1, 3 -> layer 1; 2 -> layer 2), order 1, 2, 3:
#   v = 1: 1, 1, actual 1 -> 1 [1];  v = 2: before 2, 2/2, actual 2 + 0 + 1 = 3, recall 2/3 -> 4/5 [2];
#   v = 3: before 2 (node 1), precision 2/3, actual 0 + 1 + 1 = 2, recall 1 -> 4/5 [1]
# root 1 (0, 2 -> 1; 3 -> 2), order 3, 0, 2:
#   v = 3: 1, 1, actual 3, recall 1/3 -> 1/2 [2];  v = 0: before 1, 1/2, actual 1 -> 2/3 [1];  v = 2: before 2, 2/3, actual 2, recall 1 -> 4/5 [1]
# roots 2 (order 0, 3, 1) and 3 (order 1, 2, 0): the same pattern as root 1 -> 1/2 [2], 2/3 [1], 4/5 [1]
RING4 = dict(
    edges=[(0, 1), (1, 2), (2, 3), (3, 0)], pdists=[1.0, 1.2, 3.0, 2.0, 0.7, 1.5],
    all_nodes=[moments([F(1), F(4, 5)] + [F(2, 3), F(4, 5)] * 3), moments([F(4, 5), F(1, 2), F(1, 2), F(1, 2)])],
    per_tree=[moments([F(9, 10), F(11, 15), F(11, 15), F(11, 15)]), moments([F(4, 5), F(1, 2), F(1, 2), F(1, 2)])])
assert RING4['all_nodes'] == [(F(31, 40), F(53, 4800)), (F(23, 40), F(27, 1600))] and RING4['per_tree'][0] == (F(31, 40), F(1, 192))

CASES = {'path3': PATH3, 'star4': STAR4, 'ring4': RING4}


def graph_of(case):
    import networkx as nx
    g = nx.Graph()
    g.add_nodes_from(range(1 + max(max(e) for e in case['edges'])))
    g.add_edges_from(case['edges'])
    return g


def close(got, want):
    means, var = got
    for k, (m, v) in enumerate(want):
        assert abs(float(means[k]) - float(m)) < 1e-12, (k, float(means[k]), m)
        assert abs(float(var[k]) - float(v)) < 1e-12, (k, float(var[k]), v)


@pytest.mark.parametrize('name', list(CASES))
def test_oracle_restatement_against_the_hand_derived_values(name):
    import networkx as nx
    from scipy.spatial.distance import squareform
    from oracle import ref_port as rp
    case = CASES[name]
    g = graph_of(case)
    n = g.number_of_nodes()
    hops = np.array([[nx.shortest_path_length(g, u, v) for v in range(n)] for u in range(n)], dtype=np.int64)
    dense = squareform(np.array(case['pdists']))
    deg = np.array([g.degree(u) for u in range(n)])
    close(rp.layer_f1_scores(dense, hops), case['all_nodes'])
    close(rp.layer_f1_scores(dense, hops, per_tree_average=True), case['per_tree'])
    if 'leaves_only' in case:
        close(rp.layer_f1_scores(dense, hops, min_degree=1, max_degree=1, degrees=deg), case['leaves_only'])


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
def test_gpu_evaluator_against_the_hand_derived_values(name):
    from graphembed.pyx import FastPrecision
    case = CASES[name]
    fp = FastPrecision(graph_of(case))
    pd = torch.tensor(case['pdists'], dtype=torch.float64)
    close(fp.layer_mean_f1_scores(pd), case['all_nodes'])
    close(fp.layer_mean_average_f1_scores(pd), case['per_tree'])
    if 'leaves_only' in case:
        close(fp.layer_mean_f1_scores(pd, min_degree=1, max_degree=1), case['leaves_only'])
    close(fp.layer_mean_f1_scores(pd.float()), case['all_nodes'])       # fp32 distances: the order is the same
