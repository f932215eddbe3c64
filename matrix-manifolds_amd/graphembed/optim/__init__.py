"""Riemannian optimizers over the Manifold API (fused HIP update kernels where the rule allows).

Both are graph-safe: a whole training step can be captured once and replayed (graphembed.graphed)."""
from graphembed.optim.radam import RiemannianAdam
from graphembed.optim.rsgd import RiemannianSGD

__all__ = ['RiemannianAdam', 'RiemannianSGD']
