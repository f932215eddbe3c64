"""Embedding-quality metrics — counterpart of graphembed/graphembed/metrics.py:8-96.

The correlation / distortion scores are a few reductions over the pair vectors (torch ops on the device
the vectors live on).  The mean average precision — the reference's one native component
(`pyx/impl/precision.cpp`, Cython-wrapped as `graphembed.pyx.FastPrecision`) — is a rank statistic that
needs no sort at all; it runs as a HIP kernel (`csrc/metrics.hip`, `mm_graph_average_precision`)."""
import numpy as np
import torch

from graphembed.utils import squareform1


def pearsonr(x, y):
    """Pearson correlation of two pair vectors (metrics.py:14-18)."""
    xc, yc = x - x.mean(), y - y.mean()
    return torch.dot(xc, yc) / (xc.norm() * yc.norm())


def spearmanr(x, y):
    """Spearman rank correlation (metrics.py:9-11): Pearson correlation of the average ranks."""
    import scipy.stats
    return scipy.stats.spearmanr(x.detach().cpu().numpy(), y.detach().cpu().numpy()).correlation


def average_pearsonr(mpdists, gpdists):
    """Mean over the nodes of the per-node correlation between embedding and graph distances
    (metrics.py:21-33): rows of the two dense matrices, centred by their row means."""
    m, g = squareform1(mpdists), squareform1(gpdists)
    m = m - m.mean(dim=1)   # (the reference subtracts the row-mean VECTOR without keepdim, i.e.
    g = g - g.mean(dim=1)   # column j loses the mean of row j; kept — the matrices are symmetric)
    return ((m * g).sum(dim=1) / (m.norm(dim=1) * g.norm(dim=1))).mean()


def average_distortion(mpdists, gpdists):
    """mean |d_M - d_G| / d_G (metrics.py:47-58)."""
    return ((mpdists - gpdists).abs() / gpdists).mean()


def area_under_curve(vs, step=None):
    """Trapezoid areas of per-layer score curves, one per block of `step` values (metrics.py:36-44)."""
    vs = np.asarray(vs)
    step = len(vs) if step is None else step
    usable = len(vs) // step * step
    return [0.5 * np.mean(vs[i + 1:i + step] + vs[i:i + step - 1]) for i in range(0, usable, step)]


def graph_csr(g, device):
    """CSR adjacency (int32, on `device`) of a networkx graph whose nodes are 0..n-1."""
    n = g.number_of_nodes()
    indptr = np.zeros(n + 1, dtype=np.int64)
    cols = []
    for u in range(n):
        nb = sorted(g.neighbors(u))
        cols.extend(nb)
        indptr[u + 1] = indptr[u] + len(nb)
    return (torch.from_numpy(indptr.astype(np.int32)).to(device),
            torch.from_numpy(np.asarray(cols, dtype=np.int32)).to(device))


def node_average_precision(mpdists, indptr, indices):
    """AP(u) for every node from the condensed embedding distances and a CSR adjacency — HIP kernel."""
    from graphembed import _backend as B
    B.require_gpu(mpdists, indptr, indices)
    if mpdists.dtype not in (torch.float32, torch.float64):
        mpdists = mpdists.float()
    dense = squareform1(mpdists.detach()).contiguous()
    n = dense.shape[0]
    with B.on_device(dense.device):
        ap = torch.empty(n, dtype=dense.dtype, device=dense.device)
        scratch = torch.empty(max(int(indices.numel()), 1), dtype=torch.int32, device=dense.device)
        B.lib().call('mm_graph_average_precision', B.dtype_code(dense), B.ptr(dense), n,
                     B.ptr(indptr.contiguous()), B.ptr(indices.contiguous()), B.ptr(scratch), B.ptr(ap),
                     B.stream_of(dense))
    return ap


def mean_average_precision(mpdists, g):
    """The (local) MAP of metrics.py:61-96 / FastPrecision.mean_average_precision: `mpdists` is the
    condensed (or dense n x n) matrix of embedding distances, `g` a networkx graph on nodes 0..n-1."""
    if not torch.is_tensor(mpdists):
        mpdists = torch.as_tensor(np.asarray(mpdists))
    if mpdists.ndim == 2:
        iu = torch.triu_indices(mpdists.shape[0], mpdists.shape[0], 1, device=mpdists.device)
        mpdists = mpdists[iu[0], iu[1]]
    if not mpdists.is_cuda:
        mpdists = mpdists.cuda()
    indptr, indices = graph_csr(g, mpdists.device)
    return node_average_precision(mpdists, indptr, indices).double().mean().item()
