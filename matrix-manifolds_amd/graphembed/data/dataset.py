"""Targets of the embedding problem — counterpart of graphembed/graphembed/data/dataset.py:7-30:
max-normalised SQUARED graph distances, indexable by a node subset.

The full-batch step and the multi-GPU shards only ever need the condensed pair vector (same
row-major upper-triangle order as `Manifold.pdist`), so that is what is stored; the dense n x n
matrix the reference keeps is built on first use by a node mini-batch (`dataset[indices]`,
train.py:203-213) and cached."""
import torch
from torch.utils.data import Dataset

from graphembed.utils import nnm1d2_to_n, squareform1


class GraphDataset(Dataset):

    def __init__(self, pdists):
        sq = pdists.pow(2)
        self.condensed = sq / sq.max()     # (P,)
        self._dense = None                 # (n, n), lazily

    @property
    def pdists(self):
        """Dense symmetric matrix of the targets (the reference's attribute name)."""
        if self._dense is None:
            self._dense = squareform1(self.condensed)
        return self._dense

    @property
    def device(self):
        return self.condensed.device

    def __len__(self):
        return nnm1d2_to_n(self.condensed.numel())

    def __getitem__(self, node_indices=None):
        """Pair vector of the sub-graph induced by `node_indices` (all nodes for None)."""
        if node_indices is None:
            return self.condensed
        idx = node_indices.to(self.device)
        dense = self.pdists
        if dense.is_cuda and dense.dtype in (torch.float32, torch.float64):
            from graphembed import _backend as B   # one gather kernel straight into pair-vector order
            idx = idx.to(torch.int64).contiguous()
            bs = idx.numel()
            with B.on_device(dense.device):
                out = torch.empty(bs * (bs - 1) // 2, dtype=dense.dtype, device=dense.device)
                B.lib().call('mm_pair_gather', B.dtype_code(dense), B.ptr(dense), dense.shape[0], B.ptr(idx), bs,
                             B.ptr(out), B.stream_of(dense))
            return out
        return squareform1(dense.index_select(0, idx).index_select(1, idx))

    def pairs(self, shard):
        """The slice of the full pair vector owned by a `graphembed.parallel.PairShard`."""
        return self.condensed[shard.lo:shard.hi]
