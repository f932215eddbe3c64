"""Embedding containers that call the Manifold API — counterparts of
graphembed/graphembed/modules.py:9-105 (state_dict keys `xs.k` / `scales.k` kept so
checkpoints interchange with the reference)."""
import torch
from torch.nn.functional import softplus


def _max_product_factors():
    from graphembed import _backend as B
    return B.lib().raw('mm_product_max_factors')()


class _TakeRows(torch.autograd.Function):
    """x[i] for a node minibatch; the backward is a plain index_add into zeros (the generic indexing
    backward sorts the indices first: two radix-sort kernels per step for nothing)."""

    @staticmethod
    def forward(ctx, x, i):
        ctx.save_for_backward(i)
        ctx.shape = x.shape
        return x.index_select(0, i)

    @staticmethod
    def backward(ctx, g):
        i, = ctx.saved_tensors
        return torch.zeros(ctx.shape, dtype=g.dtype, device=g.device).index_add_(0, i, g), None


def take_rows(x, i, validated=False):
    """`validated`: the caller has already run the host-side range scan over `i` (prepare_indices / normalise_indices) or
    opted out of it (BatchedObjective.check_indices = False) — the scan is one `aminmax` over the batch per call otherwise."""
    if i is None:
        return x
    if not validated:
        i = normalise_indices(i, x.shape[0])   # host-side: IndexError out of range, python-style negatives wrapped (x[idx] semantics)
    if i.dtype != torch.int64 or i.device != x.device:
        i = i.to(device=x.device, dtype=torch.int64)
    return _TakeRows.apply(x, i)


class _ProductLoss(torch.autograd.Function):
    """Objective of a product embedding with the element-wise part in ONE kernel (mm_product_loss):
    per-factor pdist forward kernels -> loss, per-factor upstream gradients and scale gradients ->
    per-factor pdist backward kernels.  Replaces ~25 framework kernels per step of
    `objective_fn(gdists, embedding.compute_dists(idx)); loss.backward()` (modules.py:84-105)."""

    @staticmethod
    def forward(ctx, target, spec, rows, manifolds, *params):
        from graphembed import _backend as B
        k = len(manifolds)
        xs, scales = params[:k], params[k:]
        B.require_gpu(*xs)
        lib = B.lib()
        kind, alpha, eps, terms = spec[:4]
        dyn = spec[4] if len(spec) > 4 else None
        dtype = xs[0].dtype
        with torch.enable_grad():
            leaves = [x.detach().requires_grad_() for x in xs]
            d2 = [man.pdist(x, squared=True, **({} if rows is None else {'rows': rows}))
                  for man, x in zip(manifolds, leaves)]
        npairs = d2[0].numel()
        tc = target.detach().to(dtype).contiguous()
        if tc.numel() != npairs:
            raise ValueError(f'target has {tc.numel()} entries, the pair range has {npairs}')
        sc = [s.detach().to(dtype).reshape(1).contiguous() for s in scales]
        dev = xs[0].device
        with B.on_device(dev):
            gs = [torch.empty(npairs, dtype=dtype, device=dev) for _ in range(k)]
            out = torch.empty(1 + k, dtype=dtype, device=dev)
            dt = B.dtype_code(xs[0])
            ws = torch.empty(lib.raw('mm_product_loss_ws_bytes')(dt, k), dtype=torch.uint8, device=dev)
            lib.call('mm_product_loss', dt, B.LOSS_STRESS if kind == 'stress' else B.LOSS_QUOTIENT, k,
                     B.ptr_array([d.detach() for d in d2]), B.ptr(tc), B.ptr_array(sc), npairs, alpha, eps,
                     terms, B.dyn_ptr(dyn, xs[0]), B.ptr_array(gs), B.ptr(out), B.ptr(ws), B.stream_of(xs[0]))
            grads = [torch.autograd.grad(d, x, g)[0] for d, x, g in zip(d2, leaves, gs)]
        ctx.grads = grads + [out[1 + i].reshape(s.shape).to(s.dtype) for i, s in enumerate(scales)]
        return out[0]

    @staticmethod
    def backward(ctx, up):
        from graphembed import _backend as B
        return (None, None, None, None) + tuple(B.take_grads(ctx, up, 'grads'))


# Largest node minibatch that is handled inside the mixed-manifold pair kernel (beyond it the step is
# arithmetic again and single factors are better served by their specialised kernels).
_SUBSET_MAX_NODES = 2048
# Workspaces of the mixed-manifold pair kernel kept per embedding (one per distinct batch shape)
_PAIR_WS_MAX = 8


def _pair_kernel_factor(man):
    """(kind, dim) of a factor the single mixed-manifold pair kernel (mm_product_pairs_loss) handles,
    else None."""
    from graphembed import _backend as B
    from graphembed.manifolds.vector import VectorManifold
    kind = getattr(man, '_kind', None)
    if isinstance(man, VectorManifold) and kind in (B.EUCLIDEAN, B.LORENTZ, B.SPHERE):
        m = getattr(man, '_m', None)
        return (kind, m) if m is not None and m <= 16 else None
    if hasattr(man, 'wmin') and hasattr(man, 'n') and not getattr(man, 'use_stein_div', False) and getattr(man, 'clamps_wide', True):
        return (B.FACTOR_SPD, man.n) if man.n in (2, 3) else None
    return None


def _pair_kernel_factors(manifolds):
    from graphembed import _backend as B
    fs = [_pair_kernel_factor(m) for m in manifolds]
    if any(f is None for f in fs) or len(fs) > 4:
        return None
    n_spd = sum(1 for k, _ in fs if k == B.FACTOR_SPD)
    if n_spd > 1 or len(fs) - n_spd > 3:
        return None
    return fs


class _ProductPairsLoss(torch.autograd.Function):
    """Objective and all gradients of a product embedding from ONE mixed-manifold pair kernel
    (mm_product_pairs_loss): every factor's squared distance, their softplus-weighted sum
    (modules.py:84-88), the loss term (objectives.py:16-45) and the gradients w.r.t. every factor's
    points and scale, without a pair vector in memory — the csphd configuration's
    Lorentz x sphere x SPD(2) step in two launches."""

    @staticmethod
    def forward(ctx, target, spec, rows, manifolds, factors, cache, subset, *params):
        """`subset` = None, or (idx, dense) for a node minibatch handled inside the kernels: `params` are
        then the FULL tables, targets are read from the dense matrix, gradients come back full-size."""
        import ctypes
        from graphembed import _backend as B
        k = len(manifolds)
        xs, scales = params[:k], params[k:]
        B.require_gpu(*xs)
        lib = B.lib()
        lkind, alpha, eps, terms = spec[:4]
        dyn = spec[4] if len(spec) > 4 else None
        dtype, dev = xs[0].dtype, xs[0].device
        n_total = xs[0].shape[0]
        n = n_total if subset is None else subset[0].numel()
        rb, re = (0, n) if rows is None else rows
        dt = B.dtype_code(xs[0])
        kinds = (ctypes.c_int * k)(*[f[0] for f in factors])
        dims = (ctypes.c_int * k)(*[f[1] for f in factors])
        wmin, wmax = 0.0, 0.0
        for man, f in zip(manifolds, factors):
            if f[0] == B.FACTOR_SPD:
                wmin, wmax = man.wmin, man.wmax
        loss_code = B.LOSS_STRESS if lkind == 'stress' else B.LOSS_QUOTIENT
        with B.on_device(dev):
            xc = [x.detach().to(dtype).contiguous() for x in xs]
            sc = [s.detach().to(dtype).reshape(1).contiguous() for s in scales]
            out = torch.empty(1 + k, dtype=dtype, device=dev)
            # The kernels leave the workspace's accumulators zero: kept across steps, it is cleared once.
            # One workspace per (dtype, device, n, factors) — a minibatch loop has one shape, its tail batch and
            # a full-batch validation loss others — and none is ever dropped while a captured graph may hold its
            # address (a replay accumulates into it and relies on finding it zero): entries touched during stream
            # capture are pinned, the oldest unpinned one goes when the table is full.
            key = (dtype, dev, n, factors)
            capturing = torch.cuda.is_current_stream_capturing()
            entry = cache.get(key) if cache is not None else None
            if entry is None:
                entry = [torch.empty(lib.raw('mm_product_pairs_ws_bytes')(dt, k, kinds, dims, n),
                                     dtype=torch.uint8, device=dev), False, False]
                if cache is not None:
                    if len(cache) >= _PAIR_WS_MAX:
                        for old in [kk for kk, e in cache.items() if not e[2]][:len(cache) - _PAIR_WS_MAX + 1]:
                            del cache[old]
                    cache[key] = entry
            ws, clean = entry[0], entry[1]
            if capturing:
                entry[2] = True   # a graph now refers to this workspace: keep it for the embedding's lifetime
            else:
                entry[1] = False  # until the call has been enqueued completely
            flags = B.WS_CLEAN if clean else 0
            if subset is None:
                tc = target.detach().to(dtype).contiguous()
                npairs = B.pair_offset(n, re) - B.pair_offset(n, rb)
                if tc.numel() != npairs:
                    raise ValueError(f'target has {tc.numel()} entries, the pair range has {npairs}')
                grads = [torch.empty_like(x) for x in xc]
                lib.call('mm_product_pairs_loss', dt, loss_code, k, kinds, dims, B.ptr_array(xc), B.ptr_array(sc),
                         B.ptr(tc), n, rb, re, alpha, eps, terms, B.dyn_ptr(dyn, xs[0]), wmin, wmax, B.ptr_array(grads), B.ptr(out),
                         B.ptr(ws), flags, B.stream_of(xs[0]))
            else:
                idx, dense = subset
                if dense.dtype != dtype or not dense.is_contiguous() or dense.shape != (n_total, n_total):
                    raise ValueError('dense targets must be a contiguous [n, n] matrix of the embedding\'s dtype')
                idx = idx.to(device=dev, dtype=torch.int64).contiguous()
                # rows outside the batch get zero gradients: ONE fill for all factors
                sizes = [x.numel() for x in xc]
                flat = torch.zeros(sum(sizes), dtype=dtype, device=dev)
                grads = [g.view(x.shape) for g, x in zip(flat.split(sizes), xc)]
                lib.call('mm_product_pairs_loss_subset', dt, loss_code, k, kinds, dims, B.ptr_array(xc),
                         B.ptr_array(sc), B.ptr(dense), n_total, B.ptr(idx), n, rb, re, alpha, eps, terms,
                         B.dyn_ptr(dyn, xs[0]), wmin, wmax, B.ptr_array(grads), B.ptr(out), B.ptr(ws), flags, B.stream_of(xs[0]))
            if not capturing:
                # clean from here on (the kernels zero what they used) — but only a call that EXECUTES proves it: a
                # call that was merely recorded leaves the flag as it found it
                entry[1] = True
        ctx.grads = [g.reshape(x.shape) for g, x in zip(grads, xs)] + \
            [out[1 + i].reshape(s.shape).to(s.dtype) for i, s in enumerate(scales)]
        return out[0]

    @staticmethod
    def backward(ctx, up):
        from graphembed import _backend as B
        return (None, None, None, None, None, None, None) + tuple(B.take_grads(ctx, up, 'grads'))


def _single_subset_factor(man):
    """(kind, dim) of a single factor whose node minibatches run inside ITS OWN pair kernel (mm_spd_pdist_loss_subset: every
    SPD(d) the library is built for; mm_vec_pdist_loss_subset: every vector manifold up to mm_vec_max_dim), else None."""
    from graphembed import _backend as B
    from graphembed.manifolds.vector import VectorManifold
    kind = getattr(man, '_kind', None)
    if isinstance(man, VectorManifold) and kind in (B.EUCLIDEAN, B.LORENTZ, B.SPHERE):
        m = getattr(man, '_m', None)
        return (kind, m) if m is not None and m <= B.lib().raw('mm_vec_max_dim')() else None
    if hasattr(man, 'wmin') and hasattr(man, 'n') and not getattr(man, 'use_stein_div', False) and getattr(man, 'clamps_wide', True):
        return (B.FACTOR_SPD, man.n) if 2 <= man.n <= B.lib().raw('mm_spd_max_dim')() else None
    return None


def distinct_in_range(indices, n):
    """Gate of every in-kernel node-minibatch route (mm_product_pairs_loss_subset, mm_{spd,vec}_pdist_loss_subset,
    mm_train_step.batch_idx): the kernels address table rows, rows of the dense target matrix and accumulator slots
    through the index vector — they read the low 32-bit word of each index and do no range check of their own — so
    the indices must be in [0, n) (the reference's `x[i]` raises IndexError otherwise, modules.py:86) and distinct
    (its indexing backward accumulates repeated rows; the per-node kernels write each gradient row once).  Slices of
    a `randperm` (train.py:206-209) always are.  HOST-side index tensors are checked here: out of range raises
    IndexError, in-range negatives (python-style) and repeats return False = "take the gather / scatter route".
    DEVICE-side index tensors cannot be checked without a synchronisation and are taken as documented (the kernels
    clamp node ids into the tables, so a bad device-side index gives wrong numbers, not a stray write)."""
    if indices.is_cuda or indices.numel() == 0:
        return True
    lo, hi = int(indices.min()), int(indices.max())
    if lo < -n or hi >= n:
        raise IndexError(f'index out of range for an embedding of {n} points: [{lo}, {hi}]')
    if lo < 0:
        return False
    return bool(torch.unique(indices).numel() == indices.numel())


def prepare_indices(indices, n, distinct=True):
    """`normalise_indices` and `distinct_in_range` in ONE host-side pass (one `aminmax`, one `unique`): returns
    `(indices, in_kernel)` — the indices as the reference's `x[idx]` reads them (IndexError out of range, python-style
    negatives wrapped) and whether the batch may take an in-kernel node-minibatch route (its nodes are distinct).  The
    training loop's entry points (BatchedObjective.forward, NativeTrainStep.__call__) call this once per step and hand
    the result down with `validated=True`; device-side tensors and empty batches pass unchecked, as documented on
    `distinct_in_range`."""
    if indices is None or indices.is_cuda or indices.numel() == 0:
        return indices, True
    lo, hi = (int(v) for v in torch.aminmax(indices))
    if lo < -n or hi >= n:
        raise IndexError(f'index out of range for an embedding of {n} points: [{lo}, {hi}]')
    if lo < 0:
        indices = torch.where(indices < 0, indices + n, indices)
    return indices, (not distinct) or bool(torch.unique(indices).numel() == indices.numel())


def normalise_indices(indices, n):
    """Host-side node indices as the reference's `x[idx]` reads them: out of range raises IndexError, python-style
    negatives are wrapped (index_select and the gather kernels take non-negative indices only).  Device-side
    tensors are returned as they are (no synchronisation on the training path)."""
    if indices is None or indices.is_cuda or indices.numel() == 0:
        return indices
    lo, hi = int(indices.min()), int(indices.max())
    if lo < -n or hi >= n:
        raise IndexError(f'index out of range for an embedding of {n} points: [{lo}, {hi}]')
    return indices if lo >= 0 else torch.where(indices < 0, indices + n, indices)


class _SingleSubsetLoss(torch.autograd.Function):
    """Objective and gradients of a node minibatch of ONE factor, evaluated inside that factor's own pair kernel
    (mm_spd_pdist_loss_subset / mm_vec_pdist_loss_subset): the index vector addresses the rows of the full table, the
    targets dense[idx[a]][idx[b]] and the gradient rows — what train.py:206-217 does with x[idx] (modules.py:86),
    dataset[idx] (data/dataset.py:19-27) and autograd's index backward, without a gather, a scatter-add or a zero fill
    (the per-node kernel writes the whole dense gradient).  SPD(4...9) and vectors wider than 16 — the sizes the
    mixed-manifold pair kernel does not take."""

    @staticmethod
    def forward(ctx, spec, rows, man, factor, cache, idx, dense, x, scale):
        from graphembed import _backend as B
        B.require_gpu(x, dense)
        lib = B.lib()
        lkind, alpha, eps, terms = spec[:4]
        dyn = spec[4] if len(spec) > 4 else None
        dtype, dev = x.dtype, x.device
        n_total, bs = x.shape[0], idx.numel()
        rb, re = (0, bs) if rows is None else rows
        dt = B.dtype_code(x)
        kind, dim = factor
        if dense.dtype != dtype or not dense.is_contiguous() or dense.shape != (n_total, n_total):
            raise ValueError('dense targets must be a contiguous [n, n] matrix of the embedding\'s dtype')
        loss_code = B.LOSS_STRESS if lkind == 'stress' else B.LOSS_QUOTIENT
        with B.on_device(dev):
            xc = x.detach().contiguous()
            sc = scale.detach().to(dtype).reshape(1).contiguous()
            ic = idx.to(device=dev, dtype=torch.int64).contiguous()
            out = torch.empty(2, dtype=dtype, device=dev)
            grad = torch.empty_like(xc)
            key = ('single', dtype, dev, n_total, kind, dim)
            ws = cache.get(key) if cache is not None else None
            if ws is None:     # kept for the embedding's lifetime: a captured graph of the step refers to it
                nbytes = (lib.raw('mm_spd_pdist_ws_bytes')(dt, n_total, dim) if kind == B.FACTOR_SPD
                          else lib.raw('mm_vec_pdist_ws_bytes')(dt, n_total, dim))
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                if cache is not None:
                    cache[key] = ws
            if kind == B.FACTOR_SPD:
                lib.call('mm_spd_pdist_loss_subset', dt, loss_code, B.ptr(xc), B.ptr(dense), B.ptr(sc), n_total, dim, B.ptr(ic), bs,
                         rb, re, alpha, eps, terms, B.dyn_ptr(dyn, x), man.wmin, man.wmax, B.ptr(out), B.ptr(grad), B.ptr(ws), 0,
                         B.stream_of(x))
            else:
                lib.call('mm_vec_pdist_loss_subset', dt, kind, loss_code, B.ptr(xc), B.ptr(dense), B.ptr(sc), n_total, dim,
                         B.ptr(ic), bs, rb, re, alpha, eps, terms, B.dyn_ptr(dyn, x), B.ptr(out), B.ptr(grad), B.ptr(ws),
                         B.stream_of(x))
        ctx.grads = [grad.reshape(x.shape), out[1].reshape(scale.shape).to(scale.dtype)]
        return out[0]

    @staticmethod
    def backward(ctx, up):
        from graphembed import _backend as B
        return (None, None, None, None, None, None, None) + tuple(B.take_grads(ctx, up, 'grads'))


class ManifoldParameter(torch.nn.Parameter):
    """A Parameter that knows the manifold it lives on (modules.py:9-23)."""

    def __new__(cls, data=None, manifold=None, requires_grad=True):
        if data is None:
            data = torch.empty(0)
        instance = torch.Tensor._make_subclass(cls, data, requires_grad)
        instance.manifold = manifold
        return instance

    def __deepcopy__(self, memo):
        # (torch's Parameter.__deepcopy__ calls type(self)(data, requires_grad): the flag would land in `manifold`;
        # the training engine deep-copies the best embedding, train.py:116-121)
        if id(self) in memo:
            return memo[id(self)]
        result = type(self)(self.data.clone(memory_format=torch.preserve_format), self.manifold, self.requires_grad)
        memo[id(self)] = result
        return result

    def proj_(self):
        self.manifold.projx(self, inplace=True)

    def __repr__(self):
        return 'Parameter on {} containing:\n'.format(self.manifold) + torch.Tensor.__repr__(self)


class ManifoldEmbedding(torch.nn.Module):
    """n points on a product of manifolds with learnable per-factor scales
    (modules.py:42-91).  Parameters are created with `manifold.rand(n)` on the
    default device, as in the reference; move with `.to(device)`."""

    def __init__(self, n, manifolds):
        super().__init__()
        self.n = n
        self.n_components = len(manifolds)
        self.pair_kernel = True  # products: use the single mixed-manifold pair kernel when it applies
        # its workspace, kept clean by the kernels themselves (one stream at a time may step an embedding)
        self._pair_ws = {}
        self.manifolds = manifolds
        self.xs = torch.nn.ParameterList(
            [ManifoldParameter(data=man.rand(n), manifold=man) for man in manifolds])
        # softplus(0.5) ~ 1 (modules.py:57-59)
        self.scales = torch.nn.ParameterList(
            [torch.nn.Parameter(torch.tensor(0.5)) for _ in manifolds])

    def _apply(self, fn, *a, **k):
        # Module.to()/cuda() rebuild Parameters; keep the manifold tag on them
        mans = [p.manifold for p in self.xs]
        out = super()._apply(fn, *a, **k)
        for p, m in zip(self.xs, mans):
            p.manifold = m
        return out

    @property
    def device(self):
        return self.xs[0].device

    @property
    def curvature_params(self):
        return self.scales

    def burnin(self, value=True):  # modules.py:36-39
        for p in self.curvature_params:
            p.requires_grad_(not value)

    @torch.no_grad()
    def perturb(self, norm):
        for x, man in zip(self.xs, self.manifolds):
            x.set_(man.retr(x, man.randvec(x, norm)))

    @torch.no_grad()
    def stabilize(self):
        for x in self.xs:
            x.proj_()

    def compute_dists(self, i=None, validated=False):
        """sum_k softplus(s_k) * pdist_k(x_k[i], squared=True) — modules.py:84-88.  (`validated`: see take_rows.)"""
        if i is not None and not validated:
            i = normalise_indices(i, self.n)   # ONE range scan for all factors
        return sum(
            softplus(s) * man.pdist(take_rows(x, i, validated=True), squared=True)
            for x, s, man in zip(self.xs, self.scales, self.manifolds))

    def fused_objective(self, objective_fn, gdists, i=None, rows=None, dense=None, params=None, validated=False, **kwargs):
        """`objective_fn(gdists, self.compute_dists(i), **kwargs)` evaluated by ONE pair kernel
        that also produces the gradients (no pair vector of distances, no element-wise passes),
        or None when the objective has no fused kernel (a loss without `fused_spec`, CPU tensors).
        Single factors run the whole pair computation in one kernel; products of up to three
        vector factors (dimension <= 16) and one SPD(2)/SPD(3) factor likewise (the mixed-manifold
        pair kernel `mm_product_pairs_loss`; `self.pair_kernel = False` turns it off); other products
        run one forward and one backward kernel per factor around a single loss kernel
        (`mm_product_loss`).  With a node minibatch `i` and the dataset's dense target matrix `dense`
        (then `gdists` may be None) the pair kernel reads rows `i` of the full tables and the targets
        `dense[i[a], i[b]]` itself and writes full-size gradients — no gather / scatter launches (the
        indices of a batch must be distinct, as slices of a `randperm` are: train.py:206-209).
        Host-side indices are range-checked, wrapped and tested for repeats HERE, once (`prepare_indices`), unless the
        caller says it has done so or has opted out (`validated=True`: BatchedObjective — whose `check_indices = False`
        skips the ~20 us of `aminmax` + `unique` per step —, NativeTrainStep): a validated batch is taken as distinct."""
        if not hasattr(objective_fn, 'fused_spec') or not self.xs[0].is_cuda:
            return None
        # `params` = (points, scales) to differentiate instead of self.xs / self.scales — views of them whose
        # backward all-reduces the gradients (graphembed.parallel.sharded_fused_objective)
        pts, scales = (list(self.xs), list(self.scales)) if params is None else params
        spec = objective_fn.fused_spec(**kwargs)
        in_kernel_batch = (i is not None and dense is not None and dense.is_cuda and self.pair_kernel
                           and dense.dtype == pts[0].dtype)
        if i is not None and not validated:
            # (host-side indices with repeats: the kernels address tables, targets and gradient rows through the index
            # vector unchecked — such a batch goes the gather / scatter way, like the reference's)
            i, distinct = prepare_indices(i, self.n, distinct=in_kernel_batch)
            in_kernel_batch = in_kernel_batch and distinct
        if in_kernel_batch and i.numel() == 0:
            # an empty batch has no pairs: a zero loss whose backward gives every parameter its (dense) zero gradient, as the
            # reference's sum over an empty pair list does
            zero = pts[0].sum() * 0
            for t in list(pts[1:]) + list(scales):
                zero = zero + t.sum() * 0
            return zero
        if in_kernel_batch:
            # node minibatch entirely inside the pair kernel: no row gathers, no target gather, no scatter-adds
            # (single factors too: at minibatch sizes a step is launches, not arithmetic)
            factors = _pair_kernel_factors(self.manifolds) if i.numel() <= _SUBSET_MAX_NODES else None
            if factors is not None:
                return _ProductPairsLoss.apply(None, spec, rows, tuple(self.manifolds), tuple(factors),
                                               self._pair_ws, (i, dense), *pts, *scales)
            if self.n_components == 1:
                # a single factor the mixed-manifold kernel does not take (SPD(4...9), vectors wider than 16): the index
                # vector goes into the factor's own pair kernel
                factor = _single_subset_factor(self.manifolds[0])
                if factor is not None:
                    return _SingleSubsetLoss.apply(spec, rows, self.manifolds[0], factor, self._pair_ws, i, dense, pts[0], scales[0])
        if self.n_components == 1 and getattr(self.manifolds[0], 'pdist_loss', None) is not None:
            if gdists is None:
                return None
            x = take_rows(pts[0], i, validated=True)
            return self.manifolds[0].pdist_loss(x, scales[0], gdists, spec, rows=rows)
        if self.n_components > _max_product_factors():
            return None
        factors = _pair_kernel_factors(self.manifolds) if self.pair_kernel else None
        if gdists is None:
            return None
        xs = [take_rows(x, i, validated=True) for x in pts]
        if factors is not None:
            return _ProductPairsLoss.apply(gdists, spec, rows, tuple(self.manifolds), tuple(factors),
                                           self._pair_ws, None, *xs, *scales)
        return _ProductLoss.apply(gdists, spec, rows, tuple(self.manifolds), *xs, *scales)

    def __len__(self):
        return self.n


class BatchedObjective(torch.nn.Module):
    """loss(dataset[idx], embedding.compute_dists(idx)) — modules.py:94-105."""

    def __init__(self, objective_fn, dataset, embedding, fused=True):
        super().__init__()
        self.objective_fn = objective_fn
        self.dataset = dataset
        self.embedding = embedding
        self.fused = fused  # use the one-pass loss+gradient kernel when the configuration has one

    # validate CPU index tensors once per step (range, python-style negatives, repeats: ~20 us of aminmax + unique); False =
    # the caller vouches for them (in range, non-negative, distinct — slices of a randperm are) and NO scan runs on any route
    check_indices = True

    def _distinct_in_range(self, indices, n):
        """`distinct_in_range` (module level) under this objective's `check_indices` switch."""
        return not self.check_indices or distinct_in_range(indices, n)

    def forward(self, indices, *args, **kwargs):
        emb = self.embedding
        distinct = True
        if self.check_indices and indices is not None:
            # the step's ONE host-side scan: IndexError like the reference's x[idx], negatives wrapped, repeats noted; every
            # route below is told so (validated=True) — with check_indices off none of them scans either
            indices, distinct = prepare_indices(indices, len(emb))
        if self.fused and not args and indices is not None and emb.xs[0].is_cuda \
                and hasattr(self.objective_fn, 'fused_spec') and hasattr(self.dataset, 'pdists'):
            dense = self.dataset.pdists   # the pair kernel gathers rows and targets itself
            if dense.is_cuda and dense.dtype == emb.xs[0].dtype and distinct:
                loss = emb.fused_objective(self.objective_fn, None, indices, dense=dense, validated=True, **kwargs)
                if loss is not None:
                    return loss
        gdists = self.dataset[indices].to(self.embedding.device)
        if self.fused and not args:
            loss = self.embedding.fused_objective(self.objective_fn, gdists, indices, validated=True, **kwargs)
            if loss is not None:
                return loss
        return self.objective_fn(gdists, self.embedding.compute_dists(indices, validated=True), *args, **kwargs)
