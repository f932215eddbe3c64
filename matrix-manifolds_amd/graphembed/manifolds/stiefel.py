from .grassmann import Stiefel  # noqa: F401  (shared implementation in grassmann.py / csrc/mat.hip)
