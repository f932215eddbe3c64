from .base import Manifold
from .spd import SymmetricPositiveDefinite
