"""Small helpers shared by the package (counterparts of graphembed/utils.py:13-145)."""
import logging
import math
import os
import time

import torch

logger = logging.getLogger(__name__)

# utils.py:13 — one epsilon for both precisions
EPS = {torch.float32: 1e-8, torch.float64: 1e-8}


def nnm1d2_to_n(m):
    """n from the pair count m = n(n-1)/2 (utils.py:27-31)."""
    n = (1 + math.isqrt(1 + 8 * m)) // 2
    if n * (n - 1) // 2 != m:
        raise AssertionError(f'{m} is not of the form n(n-1)/2')
    return n


def nnp1d2_to_n(m):
    """n from m = n(n+1)/2 (utils.py:20-24)."""
    n = (math.isqrt(1 + 8 * m) - 1) // 2
    if n * (n + 1) // 2 != m:
        raise AssertionError(f'{m} is not of the form n(n+1)/2')
    return n


def triu_mask(n, m=None, *, d=0, device=None):
    """Boolean mask of the entries on/above the d-th diagonal (utils.py:34-44)."""
    m = m or n
    r = torch.arange(n, device=device).unsqueeze(1)
    c = torch.arange(m, device=device).unsqueeze(0)
    return c - r >= d


def _to_square(x_vec, n, diag_offset):
    iu = torch.triu_indices(n, n, diag_offset, device=x_vec.device)
    sq = x_vec.new_zeros(*x_vec.shape[:-1], n, n)
    sq[..., iu[0], iu[1]] = x_vec
    sq[..., iu[1], iu[0]] = x_vec
    return sq


def squareform1(x):
    """Condensed pair vector <-> symmetric zero-diagonal matrix (utils.py:47-65)."""
    if x.ndim >= 2 and x.shape[-2] == x.shape[-1]:
        n = x.shape[-1]
        iu = torch.triu_indices(n, n, 1, device=x.device)
        return x[..., iu[0], iu[1]]
    return _to_square(x, nnm1d2_to_n(x.shape[-1]), 1)


def squareform0(x):
    """Same, including the diagonal (utils.py:68-87)."""
    if x.ndim >= 2 and x.shape[-2] == x.shape[-1]:
        n = x.shape[-1]
        iu = torch.triu_indices(n, n, 0, device=x.device)
        return x[..., iu[0], iu[1]]
    return _to_square(x, nnp1d2_to_n(x.shape[-1]), 0)


def check_mkdir(path, increment=False):
    """Create `path`; with `increment`, append _0, _1, … until unused (utils.py:102-126)."""
    if not os.path.isdir(path):
        os.makedirs(path)
        return path
    if not increment:
        logger.warning('The given path already exists (%s)', path)
        return path
    head, base = os.path.split(path)
    parts = base.split('_')
    if parts[-1].isdigit():
        base = '_'.join(parts[:-1])
    k = 0
    while os.path.isdir(os.path.join(head, f'{base}_{k}')):
        k += 1
    path = os.path.join(head, f'{base}_{k}')
    os.makedirs(path)
    logger.info('Created the directory (%s) instead', path)
    return path


class Timer:
    """`with Timer('what'):` logs the wall time of the block (utils.py:129-145)."""

    def __init__(self, msg, precision=4, loglevel=logging.DEBUG):
        self.msg, self.precision, self.loglevel = msg, precision, loglevel

    def __enter__(self):
        self.start = time.time()
        return self

    def __exit__(self, *exc):
        self.end = time.time()
        self.interval = self.end - self.start
        logger.log(self.loglevel, 'time(%s): %.*fs', self.msg, self.precision, self.interval)
