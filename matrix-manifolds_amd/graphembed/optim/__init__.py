from .rsgd import RiemannianSGD
