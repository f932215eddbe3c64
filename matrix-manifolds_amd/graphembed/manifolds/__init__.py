from .base import Manifold
from .euclidean import Euclidean
from .grassmann import Grassmann, Stiefel
from .lorentz import Lorentz
from .spd import SymmetricPositiveDefinite
from .sphere import Sphere
