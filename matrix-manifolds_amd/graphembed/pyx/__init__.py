"""`graphembed.pyx.FastPrecision` — the name under which the reference exposes its native evaluator
(pyx/precision.pyx:46-60 over pyx/impl/precision.cpp).  Here the CSR adjacency is uploaded once and the
mean average precision runs on the GPU (csrc/metrics.hip).  The layer-wise F1 curves of the reference's
class are not part of this path yet."""
import numpy as np
import torch

from graphembed.metrics import graph_csr, node_average_precision


class FastPrecision:

    def __init__(self, g, device='cuda'):
        self.n = g.number_of_nodes()
        self.n_pdists = self.n * (self.n - 1) // 2
        self.device = torch.device(device)
        self.indptr, self.indices = graph_csr(g, self.device)

    def mean_average_precision(self, mpdists):
        """Mean over the nodes of the average precision of their neighbour ranking."""
        if not torch.is_tensor(mpdists):
            mpdists = torch.from_numpy(np.ascontiguousarray(mpdists))
        if mpdists.numel() != self.n_pdists:
            raise ValueError(f'expected {self.n_pdists} pairwise distances, got {mpdists.numel()}')
        ap = node_average_precision(mpdists.to(self.device), self.indptr, self.indices)
        return ap.double().mean().item()

    def _not_yet(self, *args, **kwargs):
        raise NotImplementedError('layer-wise F1 curves (precision.cpp:300-429) are not on the GPU path yet')

    layer_mean_f1_scores = layer_mean_average_f1_scores = nodes_per_layer = _not_yet


PyFastPrecision = FastPrecision
