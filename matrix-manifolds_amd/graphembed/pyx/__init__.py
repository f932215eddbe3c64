"""`graphembed.pyx.FastPrecision` — the name under which the reference exposes its native evaluator
(pyx/precision.pyx:46-110 over pyx/impl/precision.cpp).  Here the graph side (CSR adjacency, the layers of every
shortest-path tree) is prepared once and uploaded; the mean average precision and the layer-wise F1 curves are rank
statistics counted on the GPU (csrc/metrics.hip), the row sort they walk is rocPRIM's segmented radix sort
(csrc/metrics_sort.hip).  Unweighted graphs (BFS layers) and weighted graphs — integer edge costs under the edge
attribute `weight`, as precision.pyx:117-121 detects them — whose layers are the dense ranks of the Dijkstra distances
(precision.cpp:76-92, 150-166)."""
import numpy as np
import torch

from graphembed import _backend as B
from graphembed.metrics import graph_csr, node_average_precision
from graphembed.utils import squareform1


def tree_layers(dist):
    """layers[u][v] = number of distinct graph distances from u that are smaller than d(u, v) — the layer of v in the
    shortest-path tree rooted at u (precision.cpp:150-166: a new layer starts wherever the sorted distances increase).
    For hop distances of a connected unweighted graph this is the hop distance itself."""
    n = dist.shape[0]
    layers = np.empty((n, n), dtype=np.int32)
    for u in range(n):
        _, inv = np.unique(dist[u], return_inverse=True)
        layers[u] = inv
    return layers


class FastPrecision:

    def __init__(self, g, device='cuda'):
        import networkx as nx
        from scipy.sparse.csgraph import shortest_path
        self.n = g.number_of_nodes()
        self.n_pdists = self.n * (self.n - 1) // 2
        self.device = torch.device(device)
        self.indptr, self.indices = graph_csr(g, self.device)
        edges = list(g.edges(data=True))
        self.weighted = bool(edges) and 'weight' in edges[0][2]      # precision.pyx:117-121
        if self.weighted:
            adj = nx.to_scipy_sparse_array(g, nodelist=range(self.n), weight='weight')
            if not np.all(np.equal(np.mod(adj.data, 1), 0)):
                raise ValueError('weighted graphs need integer edge costs (the reference stores them as int)')
            dist = shortest_path(adj, method='D', unweighted=False, directed=g.is_directed())
        else:
            adj = nx.to_scipy_sparse_array(g, nodelist=range(self.n))
            dist = shortest_path(adj, unweighted=True, directed=g.is_directed())
        if not np.isfinite(dist).all():
            raise ValueError('FastPrecision needs a connected graph')
        hops = tree_layers(dist) if self.weighted else dist.astype(np.int32)
        # layers of the tree rooted at u (precision.cpp:150-190); the widest tree sets the number of layers
        self.num_layers = int(hops.max()) + 1
        self._nodes_per_layer = np.bincount(hops.reshape(-1), minlength=self.num_layers)
        self.hops = torch.from_numpy(hops).to(self.device).contiguous()
        self._sort_ws = None

    # -- precision.pyx:58-60
    def mean_average_precision(self, mpdists):
        """Mean over the nodes of the average precision of their neighbour ranking."""
        ap = node_average_precision(self._pdists(mpdists, 1)[0], self.indptr, self.indices)
        return ap.double().mean().item()

    def nodes_per_layer(self):
        """Number of nodes on each layer, summed over all shortest-path trees (layer 0 = the roots)."""
        return [int(c) for c in self._nodes_per_layer]

    # -- precision.pyx:62-110
    def layer_mean_f1_scores(self, mpdists, num_pdists_sets=1, min_degree=1, max_degree=99999):
        return self._layer_f1(mpdists, num_pdists_sets, min_degree, max_degree, per_tree=False)

    def layer_mean_average_f1_scores(self, mpdists, num_pdists_sets=1):
        return self._layer_f1(mpdists, num_pdists_sets, 0, 2**31 - 1, per_tree=True)

    # ------------------------------------------------------------------------------------------------
    def _pdists(self, mpdists, sets):
        if not torch.is_tensor(mpdists):
            mpdists = torch.from_numpy(np.ascontiguousarray(mpdists))
        if mpdists.numel() != self.n_pdists * sets:
            raise ValueError(f'expected {self.n_pdists * sets} pairwise distances, got {mpdists.numel()}')
        mpdists = mpdists.to(self.device)
        if mpdists.dtype not in (torch.float32, torch.float64):
            mpdists = mpdists.float()
        return mpdists.reshape(sets, self.n_pdists)

    def _sorted_rows(self, dense):
        lib = B.lib()
        dt = B.dtype_code(dense)
        nbytes = lib.raw('mm_graph_sort_rows_ws_bytes')(dt, self.n)
        if nbytes == 0:
            raise B.BackendError(f'graphs of {self.n} nodes exceed the row sort\'s range')
        if self._sort_ws is None or self._sort_ws.numel() < nbytes or self._sort_ws.device != dense.device:
            self._sort_ws = torch.empty(nbytes, dtype=torch.uint8, device=dense.device)
        order = torch.empty(self.n, self.n, dtype=torch.int32, device=dense.device)
        lib.call('mm_graph_sort_rows', dt, B.ptr(dense), self.n, B.ptr(order), B.ptr(self._sort_ws), nbytes,
                 B.stream_of(dense))
        return order

    def _layer_f1(self, mpdists, sets, min_degree, max_degree, per_tree):
        pd = self._pdists(mpdists, sets)
        width = self.num_layers - 1
        with B.on_device(self.device):
            acc = torch.zeros(3, max(width, 1), dtype=torch.float64, device=self.device)
            for k in range(sets):
                # rows sorted by embedding distance (stable: ties by node id): mm_graph_sort_rows; the rank
                # statistics themselves are the HIP kernel
                order = self._sorted_rows(squareform1(pd[k]).contiguous())
                B.lib().call('mm_graph_layer_f1', B.ptr(order), B.ptr(self.hops), self.n, B.ptr(self.indptr),
                             int(min_degree), int(min(max_degree, 2**31 - 1)), int(per_tree), self.num_layers,
                             B.ptr(acc[0]), B.ptr(acc[1]), B.ptr(acc[2]), B.stream_of(order))
        m1, m2, cnt = acc.cpu().numpy()[:, :width]
        with np.errstate(invalid='ignore', divide='ignore'):
            means = m1 / cnt
            stds = m2 / cnt - means * means   # (the reference returns this variance under the name `stds`)
        return means, stds


PyFastPrecision = FastPrecision
