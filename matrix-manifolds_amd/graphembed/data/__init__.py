from .dataset import GraphDataset
from .graph import load_graph_pdists
