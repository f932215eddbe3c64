"""graphembed — MI355X-native implementation of the pairwise manifold-distance
training path of dalab/matrix-manifolds, behind the reference's own dotted names
(`graphembed.manifolds.*`, `graphembed.modules.*`, `graphembed.optim.*`,
`graphembed.objectives.*`).  Arithmetic runs in libmm_manifolds.so (gfx950)."""
from . import utils
from . import manifolds
from . import modules
from . import objectives
from . import optim
from . import data
from . import parallel
from . import metrics
from ._backend import unit_seed  # loss.backward(unit_seed(loss)): a backward without the `ones * grad` launches
from . import _overlay

# INTEGRATION.md option A: with the maintainer's reference checkout BEHIND this package on sys.path, the sub-modules this
# package does not define (graphembed.train, .products, .linalg, .monitor, .inference: the control plane) resolve there.
_overlay.install()

__version__ = '0.1.0'
