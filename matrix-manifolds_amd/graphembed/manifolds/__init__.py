"""The manifolds of the hot path, each routing its arithmetic to `libmm_manifolds.so` (gfx950):

  SymmetricPositiveDefinite  affine-invariant SPD(n)        csrc/spd.hip
  Lorentz, Sphere            inner-product manifolds        csrc/vec.hip, csrc/vec_gram.hip (matrix cores)
  Euclidean                  flat space                     csrc/vec.hip
  Grassmann, Stiefel         orthonormal frames             csrc/mat.hip
"""
from graphembed.manifolds.base import Manifold
from graphembed.manifolds.spd import SymmetricPositiveDefinite
from graphembed.manifolds.lorentz import Lorentz
from graphembed.manifolds.sphere import Sphere
from graphembed.manifolds.euclidean import Euclidean
from graphembed.manifolds.grassmann import Grassmann, Stiefel

__all__ = ['Manifold', 'SymmetricPositiveDefinite', 'Lorentz', 'Sphere', 'Euclidean', 'Grassmann', 'Stiefel']
