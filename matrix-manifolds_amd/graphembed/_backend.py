"""ctypes binding of libmm_manifolds.so (the C ABI in include/mm_manifolds.h).

There is exactly one compute backend: the hand-written gfx950 library.  If it is
missing or a tensor is not on the GPU, calls fail loudly — there is no CPU
fallback (use the reference itself for CPU work).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MM_MANIFOLDS_LIB points at another build of the SAME library (kernel A/B experiments)
LIB_PATH = os.environ.get('MM_MANIFOLDS_LIB') or os.path.join(os.path.dirname(_HERE), 'lib', 'libmm_manifolds.so')

MM_F32, MM_F64 = 0, 1
MM_WS_PREPARED = 1
LOSS_STRESS, LOSS_QUOTIENT = 1, 2
EUCLIDEAN, LORENTZ, SPHERE = 0, 1, 2
FACTOR_SPD = 16  # MM_FACTOR_SPD: factor kind of mm_product_pairs_loss
WS_CLEAN = 2     # MM_WS_CLEAN
SPD_EGRAD2RGRAD, SPD_EXP, SPD_RETR, SPD_LOG, SPD_PROJX, SPD_PROJU = range(6)
(FAST_SYMEIG2, FAST_SYMEIG3, FAST_CHOLESKY2, FAST_INVCHOLESKY2, FAST_SINGULAR2, FAST_DET2, FAST_DET3,
 FAST_SYMDET3) = range(8)  # MM_FAST_*

_c = ctypes
_vp, _i, _i64, _dbl, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_double, _c.c_size_t

# name -> (restype, argtypes); mirrors include/mm_manifolds.h one to one
SIGNATURES = {
    'mm_abi_version': (_i, []),
    'mm_target_arch': (_c.c_char_p, []),
    'mm_prof_enable': (_i, [_i]),
    'mm_prof_collect': (_i, [_i, _c.POINTER(_i64), _c.POINTER(_dbl)]),
    'mm_prof_clock_probe': (_i, [_vp, _i, _vp]),
    'mm_pair_offset': (_i64, [_i64, _i64]),
    'mm_shard_rows': (_i, [_i64, _i, _i, _c.POINTER(_i64), _c.POINTER(_i64)]),
    'mm_graph_layer_f1': (_i, [_vp, _vp, _i64, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'mm_graph_average_precision': (_i, [_i, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    'mm_pair_gather': (_i, [_i, _vp, _i64, _vp, _i64, _vp, _vp]),
    'mm_graph_sort_rows_ws_bytes': (_sz, [_i, _i64]),
    'mm_graph_sort_rows': (_i, [_i, _vp, _i64, _vp, _vp, _sz, _vp]),
    'mm_product_max_factors': (_i, []),
    'mm_product_loss_ws_bytes': (_sz, [_i, _i]),
    'mm_product_loss': (_i, [_i, _i, _i, _c.POINTER(_vp), _vp, _c.POINTER(_vp), _i64, _dbl, _dbl, _i, _vp,
                              _c.POINTER(_vp), _vp, _vp, _vp]),
    'mm_product_pairs_ws_bytes': (_sz, [_i, _i, _c.POINTER(_i), _c.POINTER(_i), _i64]),
    'mm_product_pairs_loss': (_i, [_i, _i, _i, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_vp), _c.POINTER(_vp),
                                    _vp, _i64, _i64, _i64, _dbl, _dbl, _i, _vp, _dbl, _dbl, _c.POINTER(_vp), _vp, _vp,
                                    _i, _vp]),
    'mm_spd_prepare': (_i, [_i, _vp, _i64, _i, _vp, _vp]),
    'mm_train_step_run': (_i, [_vp, _vp]),
    'mm_spd_fused_step_max_dim': (_i, []),
    'mm_vec_fused_step_supports': (_i, [_i, _i, _i]),
    'mm_comm_available': (_i, []),
    'mm_comm_rccl_version': (_i, []),
    'mm_comm_unique_id': (_i, [_vp]),
    'mm_comm_init': (_i, [_c.POINTER(_vp), _i, _i, _vp, _i]),
    'mm_comm_rank': (_i, [_vp]),
    'mm_comm_world': (_i, [_vp]),
    'mm_allreduce_sum': (_i, [_vp, _i, _vp, _i64, _vp]),
    'mm_comm_destroy': (_i, [_vp]),
    'mm_comm_last_error': (_c.c_char_p, []),
    'mm_vec_rsgd_multi_max': (_i, []),
    'mm_vec_rsgd_step_multi': (_i, [_i, _i, _c.POINTER(_i), _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_i64),
                                     _c.POINTER(_i), _dbl, _dbl, _i, _c.POINTER(_vp), _vp]),
    'mm_vec_radam_step': (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _dbl, _dbl, _dbl, _i, _dbl, _dbl, _i,
                                _vp, _vp]),
    'mm_spd_radam_step': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _dbl, _dbl, _dbl, _i, _dbl, _dbl, _i, _vp,
                                _vp]),
    'mm_vec_radam_step_multi': (_i, [_i, _i, _c.POINTER(_i), _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_vp),
                                      _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_i64),
                                      _c.POINTER(_i), _dbl, _dbl, _dbl, _i, _dbl, _dbl, _i, _c.POINTER(_vp), _vp]),
    'mm_product_pairs_loss_subset': (_i, [_i, _i, _i, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_vp),
                                           _c.POINTER(_vp), _vp, _i64, _vp, _i64, _i64, _i64, _dbl, _dbl, _i, _vp, _dbl,
                                           _dbl, _c.POINTER(_vp), _vp, _vp, _i, _vp]),
    'mm_vec_rsgd_momentum_step': (_i, [_i, _i, _vp, _vp, _vp, _i64, _i, _dbl, _dbl, _dbl, _dbl, _i, _vp, _vp]),
    'mm_spd_rsgd_momentum_step': (_i, [_i, _vp, _vp, _vp, _i64, _i, _dbl, _dbl, _dbl, _dbl, _i, _vp, _vp]),
    'mm_spd_max_dim': (_i, []),
    'mm_spd_pdist_ws_bytes': (_sz, [_i, _i64, _i]),
    'mm_spd_pdist_fwd': (_i, [_i, _vp, _i64, _i, _i64, _i64, _i, _dbl, _dbl, _vp, _vp, _i, _vp]),
    'mm_spd_pdist_bwd': (_i, [_i, _vp, _vp, _i64, _i, _i64, _i64, _i, _dbl, _dbl, _vp, _vp, _i, _vp]),
    'mm_spd_pdist_loss': (_i, [_i, _i, _vp, _vp, _vp, _i64, _i, _i64, _i64, _dbl, _dbl, _i, _vp, _dbl, _dbl, _vp, _vp,
                                _vp, _i, _vp]),
    'mm_spd_pdist_loss_subset': (_i, [_i, _i, _vp, _vp, _vp, _i64, _i, _vp, _i64, _i64, _i64, _dbl, _dbl, _i, _vp, _dbl, _dbl,
                                       _vp, _vp, _vp, _i, _vp]),
    'mm_vec_pdist_loss_subset': (_i, [_i, _i, _i, _vp, _vp, _vp, _i64, _i, _vp, _i64, _i64, _i64, _dbl, _dbl, _i, _vp, _vp, _vp,
                                       _vp, _vp]),
    'mm_spd_stein_pdiv_fwd': (_i, [_i, _vp, _i64, _i, _i64, _i64, _i, _dbl, _vp, _vp, _i, _vp]),
    'mm_spd_stein_pdiv_bwd': (_i, [_i, _vp, _vp, _i64, _i, _i64, _i64, _i, _dbl, _vp, _vp, _i, _vp]),
    'mm_spd_stein_div': (_i, [_i, _vp, _vp, _vp, _i64, _i, _i, _dbl, _vp, _vp, _vp, _vp]),
    'mm_spd_status': (_i, [_vp, _i64, _c.POINTER(_i), _vp]),
    'mm_spd_dist_fwd': (_i, [_i, _vp, _vp, _i64, _i, _i, _dbl, _dbl, _vp, _vp]),
    'mm_spd_dist_bwd': (_i, [_i, _vp, _vp, _vp, _i64, _i, _i, _dbl, _dbl, _vp, _vp, _vp]),
    'mm_spd_map': (_i, [_i, _i, _vp, _vp, _i64, _i, _dbl, _dbl, _vp, _vp]),
    'mm_spd_eigvalsh': (_i, [_i, _vp, _i64, _i, _vp, _vp]),
    'mm_fast_fwd': (_i, [_i, _i, _vp, _i64, _dbl, _vp, _vp, _vp]),
    'mm_fast_bwd': (_i, [_i, _i, _vp, _vp, _vp, _i64, _dbl, _vp, _vp]),
    'mm_spd_norm': (_i, [_i, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    'mm_spd_rsgd_step': (_i, [_i, _vp, _vp, _i64, _i, _dbl, _dbl, _i, _vp, _vp]),
    'mm_vec_max_dim': (_i, []),
    'mm_vec_pdist_ws_bytes': (_sz, [_i, _i64, _i]),
    'mm_vec_pdist_fwd': (_i, [_i, _i, _vp, _i64, _i, _i64, _i64, _i, _vp, _vp]),
    'mm_vec_pdist_fwd_gram': (_i, [_i, _i, _vp, _i64, _i, _i64, _i64, _i, _vp, _vp]),
    'mm_vec_pdist_bwd': (_i, [_i, _i, _vp, _vp, _i64, _i, _i64, _i64, _i, _vp, _vp, _vp]),
    'mm_vec_pdist_bwd_gram': (_i, [_i, _i, _vp, _vp, _i64, _i, _i64, _i64, _i, _vp, _vp]),
    'mm_vec_pdist_loss': (_i, [_i, _i, _i, _vp, _vp, _vp, _i64, _i, _i64, _i64, _dbl, _dbl, _i, _vp, _vp, _vp, _vp, _vp]),
    'mm_vec_dist': (_i, [_i, _i, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    'mm_vec_map': (_i, [_i, _i, _i, _vp, _vp, _vp, _i64, _i, _vp, _vp]),
    'mm_vec_norm': (_i, [_i, _i, _vp, _i64, _i, _i, _vp, _vp]),
    'mm_vec_rsgd_step': (_i, [_i, _i, _vp, _vp, _i64, _i, _dbl, _dbl, _i, _vp, _vp]),
    'mm_mat_max_rows': (_i, []),
    'mm_mat_max_cols': (_i, []),
    'mm_mat_map': (_i, [_i, _i, _i, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    'mm_grass_dist': (_i, [_i, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'mm_grass_pdist_ws_bytes': (_sz, [_i, _i64, _i, _i]),
    'mm_grass_pdist_fwd': (_i, [_i, _vp, _i64, _i, _i, _i64, _i64, _i, _vp, _vp]),
    'mm_grass_pdist_bwd': (_i, [_i, _vp, _vp, _i64, _i, _i, _i64, _i64, _i, _vp, _vp, _vp]),
}
GRASSMANN, STIEFEL = 0, 1
MAT_PROJU, MAT_PROJX, MAT_RETR_SVD, MAT_RETR_QR, MAT_EXP, MAT_LOG = range(6)
VEC_EGRAD2RGRAD, VEC_PROJU, VEC_EXP, VEC_RETR, VEC_PROJX, VEC_TRANSP, VEC_LOG = range(7)


class BackendError(RuntimeError):
    pass


def dtype_code(t):
    if t.dtype == torch.float32:
        return MM_F32
    if t.dtype == torch.float64:
        return MM_F64
    raise TypeError(f'matrix-manifolds_amd kernels exist for float32/float64, got {t.dtype}')


class HipLibrary:
    """Thin, typed view of the shared library. Every call checks the return code."""

    def __init__(self, path=LIB_PATH):
        if not os.path.isfile(path):
            raise BackendError(
                f'{path} not found: build it with `python __graft_entry__.py` (hipcc, gfx950). '
                'There is no CPU fallback.')
        self.path = path
        self._lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self._lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        arch = self._lib.mm_target_arch().decode()
        if arch != 'gfx950':
            raise BackendError(f'library built for {arch}, expected gfx950')

    def call(self, name, *args):
        rc = getattr(self._lib, name)(*args)
        if rc != 0:
            if rc == -3:
                kind = 'collective: ' + self._lib.mm_comm_last_error().decode(errors='replace')
            else:
                kind = {-1: 'invalid argument', -2: 'unsupported size/dtype'}.get(rc, f'hipError_t {rc}')
            raise BackendError(f'{name} failed: {kind}')

    def raw(self, name):
        return getattr(self._lib, name)


_instance = None


def lib():
    """The process-wide library handle (loaded on first use; raises if absent)."""
    global _instance
    if _instance is None:
        _instance = HipLibrary()
    return _instance


_autograd = False     # False: not looked up yet; None: unavailable


def autograd_ext():
    """The C++ autograd nodes of `pdist` (csrc_torch/mm_autograd.cpp, built in-tree as lib/_mm_autograd.so): forward and
    backward are one C-ABI call each, issued from C++ — the engine does not re-enter Python for the backward.  None if the
    module is not built or MM_PY_AUTOGRAD=1: the torch.autograd.Function classes of graphembed.manifolds issue the same
    two calls from Python (same kernels, ~80 us more host time per forward + backward)."""
    global _autograd
    if _autograd is False:
        _autograd = None
        path = os.path.join(os.path.dirname(_HERE), 'lib', '_mm_autograd.so')
        if os.environ.get('MM_PY_AUTOGRAD', '') != '1' and os.path.isfile(path):
            import importlib.util
            lib()                      # (the HIP library: raises if absent — there is no CPU fallback)
            try:
                stamp = os.path.join(os.path.dirname(path), '_mm_autograd.stamp')
                built_for = open(stamp).read().splitlines()[0] if os.path.isfile(stamp) else None
                if built_for != f'torch {torch.__version__}':   # (csrc_torch/build.py writes it; a binary for another torch loads and then misbehaves)
                    raise ImportError(f'built for {built_for}, this interpreter has torch {torch.__version__}')
                spec = importlib.util.spec_from_file_location('_mm_autograd', path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                mod.init(LIB_PATH)
                _autograd = mod
            except (ImportError, OSError, RuntimeError) as e:   # built against another torch: host the calls from Python
                import warnings
                warnings.warn(f'{path} does not load ({e}); rebuild it with `python __graft_entry__.py`. pdist stays on the '
                              'HIP kernels, hosted by the Python autograd classes')
    return _autograd


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise BackendError(
                'matrix-manifolds_amd runs on MI355X only: got a CPU tensor. Move the embedding to '
                "'cuda' (torch-ROCm device); CPU execution is the reference's job.")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def dyn_ptr(dyn, like):
    """Pointer of the device-resident loss schedule {alpha, eps} (QuotientLoss.on_device) — it is dereferenced by
    the kernels launched on `like`'s device, so it must live there."""
    if dyn is None:
        return None
    if dyn.device != like.device or dyn.dtype != torch.float64 or dyn.numel() < 2:
        raise BackendError(f'loss schedule lives on {dyn.device} ({dyn.dtype}); the embedding is on {like.device}: '
                           'call objective_fn.on_device(embedding.device)')
    return ctypes.c_void_p(dyn.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_of(t):
    """The current HIP stream of the tensor's device as a `hipStream_t` (the raw-handle query costs
    ~0.3 us; `torch.cuda.current_stream()` builds a Stream object, ~5 us per call)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(t.device.index if t.device.index is not None
                                           else torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class on_device:
    """`with torch.cuda.device(d)` that costs nothing when `d` already is the current device (the
    usual case: one process per GPU); allocations and launches inside go to `d`."""

    __slots__ = ('ctx', )

    def __init__(self, device):
        idx = device.index
        self.ctx = None if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


# ---- pair-list geometry (pure host arithmetic; usable without a GPU) ----------
def pair_offset(n, row):
    return row * (2 * n - row - 1) // 2


def shard_rows(n, world, rank):
    """Row range of shard ``rank``: contiguous, balanced by the COST of the pair kernel — a pair of a row of L pairs counts
    1 + L / K (K = 400 000: the row-side atomics of long rows contend; csrc/common.hip has the measurement).

    Same rule as ``mm_shard_rows`` (csrc/common.hip), restated in Python (exact integers) so that host logic can be
    tested without the library.  MM_SHARD_K overrides K in both (0: balance pair counts)."""
    K = int(os.environ.get('MM_SHARD_K', 400000))

    def cost_before(row):
        m1, m0 = n - 1, n - 1 - row
        sum1 = m1 * (m1 + 1) // 2 - m0 * (m0 + 1) // 2
        if K <= 0:
            return sum1
        return sum1 * K + m1 * (m1 + 1) * (2 * m1 + 1) // 6 - m0 * (m0 + 1) * (2 * m0 + 1) // 6

    def first_row_at_or_after(target):
        lo, hi = 0, n
        while lo < hi:
            mid = (lo + hi) // 2
            if cost_before(mid) >= target:
                hi = mid
            else:
                lo = mid + 1
        return lo

    total = cost_before(n) if n > 0 else 0
    rb = 0 if rank == 0 else first_row_at_or_after(total * rank // world)
    re = n if rank == world - 1 else first_row_at_or_after(total * (rank + 1) // world)
    return rb, re


def ptr_array(tensors):
    """Host array of device pointers (the `const void* const*` arguments of the C ABI)."""
    import ctypes
    arr = (ctypes.c_void_p * len(tensors))()
    for k, t in enumerate(tensors):
        arr[k] = t.data_ptr()
    return arr


# ---- the seed of a backward pass -------------------------------------------------------------------
# `loss.backward()` materialises ones_like(loss) and the fused objective Functions then multiply every
# stored gradient by it: two launches of a step that is ~10 launches long.  `unit_seed(loss)` is a cached,
# read-only tensor holding 1; a backward seeded with it (`loss.backward(unit_seed(loss))`, what
# GraphedTrainStep does) is recognised by its storage address and the multiplication is skipped.
_unit_seeds = {}


def unit_seed(like):
    """The cached 0-dim tensor 1 of `like`'s dtype and device.  Never write to it."""
    import torch
    key = (like.dtype, like.device)
    t = _unit_seeds.get(key)
    if t is None:
        t = torch.ones((), dtype=like.dtype, device=like.device)
        _unit_seeds[key] = t
    return t


def is_unit_seed(up):
    seed = _unit_seeds.get((up.dtype, up.device))
    return seed is not None and up.numel() == 1 and up.data_ptr() == seed.data_ptr()


def take_grads(ctx, up, *names):
    """The gradients a fused objective Function stored on `ctx` under `names`, times `up`.  Seeded
    with a unit seed they are handed over as they are AND released from `ctx`, so that autograd can
    adopt them as `.grad` without a copy (it clones gradients somebody else still references); such
    a backward can therefore run once."""
    vals = []
    for name in names:
        v = getattr(ctx, name)
        vals.extend(v) if isinstance(v, (list, tuple)) else vals.append(v)
    if is_unit_seed(up):
        if getattr(ctx, '_taken', False):
            raise RuntimeError('the gradients of this fused objective were handed over by an earlier '
                               'backward seeded with unit_seed(); seed with a plain tensor to backward twice')
        ctx._taken = True
        for name in names:
            setattr(ctx, name, None)
        return vals
    if getattr(ctx, '_taken', False):
        raise RuntimeError('the gradients of this fused objective were handed over by an earlier backward')
    return scale_grads(vals, up)


def scale_grads(grads, up):
    """[g * up for g in grads] (None entries kept)."""
    import torch
    live = [g for g in grads if g is not None]
    try:  # one multi-tensor launch
        scaled = list(torch._foreach_mul(live, up))
    except (RuntimeError, TypeError):
        scaled = [g * up for g in live]
    it = iter(scaled)
    return [None if g is None else next(it) for g in grads]
