from .rsgd import RiemannianSGD
from .radam import RiemannianAdam
