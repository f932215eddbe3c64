"""Golden values of the reference's metrics (development container only).
    PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0 python tests/golden/gen_golden_metrics.py
py_mean_average_precision, average_distortion, pearsonr, average_pearsonr, area_under_curve of
graphembed/graphembed/metrics.py on seeded Erdos-Renyi graphs; output tests/golden/metrics.npz."""
import os
import sys

import networkx as nx
import numpy as np
import torch
from scipy.spatial.distance import squareform

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from graphembed import metrics as M  # noqa: E402


def main():
    out = {}
    rng = np.random.default_rng(7)
    for n, p in ((50, 0.1), (100, 0.05), (100, 0.5), (300, 0.02)):
        for seed in (0, 1):
            g = nx.erdos_renyi_graph(n, p, seed=seed)
            g = g.subgraph(max(nx.connected_components(g), key=len)).copy()
            g = nx.convert_node_labels_to_integers(g)
            m = g.number_of_nodes()
            pd = rng.random(m * (m - 1) // 2).astype(np.float32)
            tag = f'er{n}_{p}_{seed}'
            out[f'{tag}/edges'] = np.asarray(g.edges(), dtype=np.int32)
            out[f'{tag}/n'] = np.int64(m)
            out[f'{tag}/pdists'] = pd
            out[f'{tag}/map'] = np.float64(M.py_mean_average_precision(squareform(pd), g))
            gd = rng.random(pd.shape).astype(np.float64) + 0.5
            md = gd + 0.1 * rng.standard_normal(pd.shape)
            out[f'{tag}/gd'], out[f'{tag}/md'] = gd, md
            out[f'{tag}/distortion'] = np.float64(M.average_distortion(torch.from_numpy(md), torch.from_numpy(gd)))
            out[f'{tag}/pearsonr'] = np.float64(M.pearsonr(torch.from_numpy(md), torch.from_numpy(gd)))
            out[f'{tag}/avg_pearsonr'] = np.float64(M.average_pearsonr(torch.from_numpy(md), torch.from_numpy(gd)))
    vs = rng.random(12)
    out['auc/vs'] = vs
    out['auc/full'] = np.asarray(M.area_under_curve(vs))
    out['auc/step4'] = np.asarray(M.area_under_curve(vs, 4))
    np.savez_compressed(os.path.join(HERE, 'metrics.npz'), **out)
    print(len(out), 'arrays')


if __name__ == '__main__':
    main()
