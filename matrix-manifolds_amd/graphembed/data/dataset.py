"""Targets of the embedding problem — counterpart of graphembed/graphembed/data/dataset.py:7-30.
Stores max-normalised SQUARED graph distances; indexing by a node subset returns the
row-major upper-triangle pair vector of that subset (same order as Manifold.pdist)."""
import torch
from torch.utils.data import Dataset

from graphembed.utils import squareform1


class GraphDataset(Dataset):

    def __init__(self, pdists):
        pdists = pdists.pow(2)
        pdists = pdists / pdists.max()
        self.condensed = pdists            # (P,) — what full-batch steps and shards read
        self.pdists = squareform1(pdists)  # dense (n,n) for node mini-batches

    @property
    def device(self):
        return self.pdists.device

    def __getitem__(self, node_indices=None):
        if node_indices is None:
            return self.condensed
        node_indices = node_indices.to(self.device)
        sub = self.pdists[node_indices][:, node_indices]
        return squareform1(sub)

    def pairs(self, shard):
        """The slice of the full pair vector owned by a `graphembed.parallel.PairShard`."""
        return self.condensed[shard.lo:shard.hi]

    def __len__(self):
        return len(self.pdists)
