"""Targets of the embedding problem: `GraphDataset` (squared, max-normalised shortest-path distances in
pair-vector order) and `load_graph_pdists` (edge list / cached distance matrix -> pair vector)."""
from graphembed.data.graph import load_graph_pdists
from graphembed.data.dataset import GraphDataset

__all__ = ['GraphDataset', 'load_graph_pdists']
