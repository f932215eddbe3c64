"""A full-batch training step issued by ONE call into the HIP library (`mm_train_step_run`, csrc/step.hip): the fused
objective kernel of the embedding followed by the fused optimizer kernels of all its parameters.

The reference's loop body (graphembed/graphembed/train.py:198-222)

    loss = objective(None, epoch=epoch, alpha=alpha); optimizer.zero_grad(); loss.backward(); optimizer.step()

costs 130-400 us of Python per step on this package's classes (tensor wrappers, autograd nodes, ~15 foreign-function
calls) for ~100 us of kernels.  Callers that replay a captured HIP graph (`graphembed.graphed.GraphedTrainStep`) do
not pay that; callers that cannot capture — changing shapes, host logic between steps, debuggers — use this:

    step = NativeTrainStep(embedding, objective_fn, targets, [opt_points, opt_scales])
    for epoch in range(n_epochs):
        loss = step(epoch=epoch, alpha=1.0)      # device scalar, the loss before the update

The optimizers stay the owners of hyper-parameters and state (learning-rate schedulers, `state_dict`, momentum buffers
and Adam moments are the optimizers' own tensors); `p.grad` of every parameter is a persistent buffer that the step
overwrites.  Supported: what the fused objective kernels support — one SPD or vector factor, or a product of up to
three vector factors (dimension <= 16) and one SPD(2)/SPD(3) factor — with StressLoss / QuotientLoss, RiemannianSGD
(with or without momentum) and RiemannianAdam.  Anything else raises `ValueError` at construction.

Every single-GPU step is TWO launches: the pair kernel of the embedding and one per-point kernel (gradient from the pair
kernel's sums, loss record, optimizer rule of every parameter, momentum-free RSGD scales).  A single SPD or vector factor
also keeps what its pair kernel reads up to date itself (that kernel writes the per-node tables / the zero-padded copy of
the new points), so consecutive steps skip the preparation launch; an edit of the points from outside
(`stabilize`, a manual in-place operation) is noticed through the tensor's storage / version and re-prepares.  A step that
was CAPTURED into a HIP graph replays the launch sequence it was recorded with: after editing the points outside the
graph, issue one eager `step()` (it prepares again) before replaying.

Multi-GPU (one process per GPU): `NativeTrainStep(..., shard=PairShard(n), comm=Communicator...)` evaluates this rank's
rows of the pair list, all-reduces {gradients, loss, scale gradients} ONCE through the library's RCCL communicator between
the objective and the optimizer kernels — inside the same C call, on the same stream, no host round trip — and applies
the identical update on every rank (the replacement of train.py:107-109, torch.nn.DataParallel).  `targets` may be the
full pair vector or this rank's slice."""
import ctypes

import torch

from graphembed import _backend as B

_c = ctypes


class _StepParam(_c.Structure):
    _fields_ = [('kind', _c.c_int), ('dim', _c.c_int), ('count', _c.c_int64), ('x', _c.c_void_p), ('grad', _c.c_void_p),
                ('optimizer', _c.c_int), ('lr', _c.c_double), ('momentum', _c.c_double), ('dampening', _c.c_double),
                ('max_grad_norm', _c.c_double), ('beta1', _c.c_double), ('beta2', _c.c_double), ('adam_eps', _c.c_double),
                ('nc', _c.c_int), ('exact', _c.c_int), ('state0', _c.c_void_p), ('state1', _c.c_void_p),
                ('step', _c.c_void_p), ('ticket', _c.c_void_p)]


class _TrainStep(_c.Structure):
    _fields_ = [('struct_size', _c.c_size_t), ('dtype', _c.c_int), ('loss_kind', _c.c_int), ('terms', _c.c_int), ('alpha', _c.c_double),
                ('eps', _c.c_double), ('loss_params', _c.c_void_p), ('wmin', _c.c_double), ('wmax', _c.c_double),
                ('n', _c.c_int64), ('nf', _c.c_int), ('points', _StepParam * 4), ('scales', _StepParam * 4),
                ('target', _c.c_void_p), ('loss_out', _c.c_void_p), ('ws', _c.c_void_p), ('ws_flags', _c.c_int),
                ('row_begin', _c.c_int64), ('row_end', _c.c_int64), ('comm', _c.c_void_p), ('reduce_buf', _c.c_void_p),
                ('reduce_count', _c.c_int64), ('batch_idx', _c.c_void_p), ('batch', _c.c_int64)]

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.struct_size = _c.sizeof(type(self))   # the struct is versioned by its size (mm_abi_version() >= 4)


OPT_NONE, OPT_RSGD, OPT_RADAM = -1, 0, 1


def _factor_of(man):
    """(kind, dim) of a manifold for the C ABI."""
    from graphembed.manifolds.vector import VectorManifold
    if isinstance(man, VectorManifold):
        return man._kind, man._m
    if hasattr(man, 'wmin') and hasattr(man, 'n') and not getattr(man, 'use_stein_div', False) and getattr(man, 'clamps_wide', True):
        return B.FACTOR_SPD, man.n
    raise ValueError(f'no fused training step for {man}')


class NativeTrainStep:

    check_indices = True   # distinctness of HOST-side minibatch indices (a torch.unique per step); the range check always runs

    def __init__(self, embedding, objective_fn, targets, optimizers, shard=None, comm=None, dense=None):
        """`dense`: the dataset's dense [n, n] matrix of squared graph distances (GraphDataset.pdists) — enables node
        minibatches of a single factor, `step(indices=idx, ...)` (train.py:198-222 with batch_size set): the index vector goes
        into the pair kernel, the optimizer kernel still steps all n points (zero gradient outside the batch, as the
        reference's dense x.grad has it).  `targets` may then be None (minibatch steps only)."""
        from graphembed.modules import _pair_kernel_factors
        from graphembed.optim import RiemannianAdam, RiemannianSGD
        if not hasattr(objective_fn, 'fused_spec'):
            raise ValueError(f'{objective_fn} has no fused objective kernel')
        xs, scales = list(embedding.xs), list(embedding.scales)
        if not xs or not all(x.is_cuda for x in xs):
            raise ValueError('NativeTrainStep needs the embedding in GPU memory')
        k = len(xs)
        dtype, dev = xs[0].dtype, xs[0].device
        if dtype not in (torch.float32, torch.float64) or any(p.dtype != dtype or not p.is_contiguous() for p in xs + scales):
            raise ValueError('parameters must be contiguous tensors of one floating-point dtype')
        factors = [_factor_of(m) for m in embedding.manifolds]
        if k > 1:
            if k > 4 or _pair_kernel_factors(embedding.manifolds) is None:
                raise ValueError('products are served by the mixed-manifold pair kernel only: at most three vector '
                                 'factors of dimension <= 16 and one SPD(2) / SPD(3) factor')
        else:
            kind, dim = factors[0]
            cap = B.lib().raw('mm_spd_max_dim')() if kind == B.FACTOR_SPD else B.lib().raw('mm_vec_max_dim')()
            if dim > cap:
                raise ValueError(f'dimension {dim} exceeds the kernels\' range ({cap})')
        n = xs[0].shape[0]
        npairs = n * (n - 1) // 2
        if dense is not None:
            if k != 1:
                raise ValueError('node minibatches inside the one-call step exist for a single factor')
            if not dense.is_cuda or dense.dtype != dtype or tuple(dense.shape) != (n, n) or not dense.is_contiguous():
                raise ValueError('dense must be a contiguous [n, n] GPU matrix of the embedding\'s dtype')
        self.dense, self._idx = dense, None
        if targets is None:
            if dense is None:
                raise ValueError('targets (pair vector) or dense (n x n matrix) is needed')
            targets = torch.empty(0, dtype=dtype, device=dev)
            npairs = 0
        if shard is not None:
            if shard.n != n:
                raise ValueError(f'the shard is cut for {shard.n} points, the embedding has {n}')
            if targets.numel() == npairs:
                targets = shard.slice(targets)
            npairs = shard.num_pairs
        if comm is not None and (shard is None or comm.world != shard.world or comm.rank != shard.rank):
            raise ValueError('a communicator needs the matching PairShard (same world size and rank)')
        if shard is not None and shard.world > 1 and comm is None:
            # (the collective sits INSIDE mm_train_step_run: without a communicator every rank would step on its own partial
            # gradient.  Ranks that share a GPU cannot build an RCCL communicator — they use parallel.sharded_fused_objective,
            # whose all-reduce goes through torch.distributed)
            raise ValueError('a step sharded over several ranks needs the RCCL communicator (graphembed.comm.Communicator)')
        if targets.numel() != npairs:
            raise ValueError(f'targets has {targets.numel()} entries, the step covers {npairs} pairs')
        self.shard, self.comm = shard, comm
        self.embedding, self.objective_fn, self.optimizers = embedding, objective_fn, list(optimizers)
        self.n, self.k, self.dtype, self.device = n, k, dtype, dev
        self._opt_of = {}
        for o in self.optimizers:
            if not isinstance(o, (RiemannianSGD, RiemannianAdam)):
                raise ValueError(f'{type(o).__name__}: only RiemannianSGD / RiemannianAdam have fused update kernels')
            for g in o.param_groups:
                for p in g['params']:
                    self._opt_of[id(p)] = (o, g)
        lib = B.lib()
        dt = B.dtype_code(xs[0])
        with B.on_device(dev):
            self.target = targets.detach().to(device=dev, dtype=dtype).contiguous()
            # ONE allocation for everything the collective of a sharded step sums: the point gradients of every factor,
            # then {loss, scale gradients} — the message of mm_allreduce_sum (mm_train_step.reduce_buf)
            sizes = [x.numel() for x in xs]
            self.flat = torch.zeros(sum(sizes) + 1 + k, dtype=dtype, device=dev)
            parts = torch.split(self.flat, sizes + [1 + k])
            self.grads = [g.view_as(x) for g, x in zip(parts[:k], xs)]
            self.loss_out = parts[k]
            kinds = (_c.c_int * k)(*[f[0] for f in factors])
            dims = (_c.c_int * k)(*[f[1] for f in factors])
            if k > 1:
                nbytes = lib.raw('mm_product_pairs_ws_bytes')(dt, k, kinds, dims, n)
            elif factors[0][0] == B.FACTOR_SPD:
                nbytes = lib.raw('mm_spd_pdist_ws_bytes')(dt, n, factors[0][1])
            else:
                nbytes = lib.raw('mm_vec_pdist_ws_bytes')(dt, n, factors[0][1])
            self.ws = torch.zeros(max(int(nbytes), 64), dtype=torch.uint8, device=dev)
        for x, g in zip(xs, self.grads):
            x.grad = g                                        # persistent: the step overwrites it
        for i, s in enumerate(scales):
            s.grad = self.loss_out[1 + i].view(s.shape)        # the scale gradients live in loss_out
        self._desc = d = _TrainStep()
        d.dtype, d.n, d.nf = dt, n, k
        d.wmin, d.wmax = 1e-8, 1e8
        for man in embedding.manifolds:
            if hasattr(man, 'wmin'):
                d.wmin, d.wmax = man.wmin, man.wmax
        d.target, d.loss_out, d.ws = self.target.data_ptr(), self.loss_out.data_ptr(), self.ws.data_ptr()
        d.ws_flags = 0
        if shard is not None:
            d.row_begin, d.row_end = shard.rows
            if shard.rows[1] <= 0:      # (row_end <= 0 means n in the ABI: an empty shard at row 0 is rows [0, 0) of 0 pairs)
                d.row_begin, d.row_end = n, n
        if comm is not None:
            d.comm, d.reduce_buf, d.reduce_count = comm.handle.value, self.flat.data_ptr(), self.flat.numel()
        for i, (x, f) in enumerate(zip(xs, factors)):
            q = d.points[i]
            q.kind, q.dim, q.count, q.x, q.grad = f[0], f[1], n, x.data_ptr(), self.grads[i].data_ptr()
        for i, s in enumerate(scales):
            q = d.scales[i]
            q.kind, q.dim, q.count = B.EUCLIDEAN, 1, 1
            q.x = s.data_ptr()
            q.grad = None
        self._keep = []     # optimizer state tensors referenced by the descriptor
        self._params = xs + scales
        self._slots = [d.points[i] for i in range(k)] + [d.scales[i] for i in range(k)]
        self._trainable_scale = [True] * k
        # a single SPD factor: the optimizer kernel of a step also writes the per-node tables of the NEW points, so the next
        # step skips the preparation launch (MM_WS_PREPARED) — as long as nobody else touched the points in between
        # (and so does the per-point kernel of a single vector factor: the zero-padded copy of the new points)
        self._single_spd = k == 1 and factors[0][0] == B.FACTOR_SPD
        if self._single_spd:
            self._prepared_single = factors[0][1] <= lib.raw('mm_spd_fused_step_max_dim')()
        else:
            self._prepared_single = k == 1 and bool(lib.raw('mm_vec_fused_step_supports')(dt, factors[0][0], factors[0][1]))
        self._tables_of = None

    # ------------------------------------------------------------------------------------------------------------
    def _bind_optimizer(self, p, q):
        """Hyper-parameters (read every step: schedulers change them) and state pointers of parameter `p`."""
        from graphembed.optim import RiemannianAdam
        from graphembed.utils import EPS
        entry = self._opt_of.get(id(p))
        if entry is None or not p.requires_grad:
            return False
        o, g = entry
        clip = g['max_grad_norm']
        q.lr, q.max_grad_norm, q.exact = float(g['lr']), (-1.0 if clip is None else float(clip)), int(bool(g['exact']))
        if isinstance(o, RiemannianAdam):
            m, v, t = o._moments(p)
            ticket = o._ticket(p)
            q.optimizer = OPT_RADAM
            q.beta1 = float(g['betas'][0])
            q.beta2 = float(g['betas'][1] if g['betas'][1] is not None else 0.0)
            q.nc, q.adam_eps = int(bool(g['nc'])), float(EPS[p.dtype])
            q.state0, q.state1, q.step, q.ticket = m.data_ptr(), v.data_ptr(), t.data_ptr(), ticket.data_ptr()
        else:
            q.optimizer = OPT_RSGD
            q.momentum, q.dampening = float(g['momentum']), float(g['dampening'])
            if g['momentum'] != 0:
                st = o.state[p]
                if 'momentum_buffer' not in st:
                    # the reference starts the buffer as a clone of the first EUCLIDEAN gradient (rsgd.py:53-54):
                    # that gradient exists only after the objective of this step — the first step runs unfused
                    return None
                q.state0 = st['momentum_buffer'].data_ptr()
            else:
                q.state0 = None
        return True

    def __call__(self, indices=None, **objective_kwargs):
        spec = self.objective_fn.fused_spec(**objective_kwargs)
        d = self._desc
        if indices is not None:
            if self.dense is None:
                raise ValueError('a minibatch step needs the dense target matrix: NativeTrainStep(..., dense=dataset.pdists)')
            if self.shard is not None:
                raise ValueError('minibatch steps of NativeTrainStep run on one GPU')
            # the kernels behind mm_train_step.batch_idx address table rows, dense-target rows and accumulator slots through
            # the index vector unchecked: host-side indices are validated here as BatchedObjective does (IndexError out of range
            # like the reference's x[idx], python-style negatives wrapped); device-side ones cannot be without a synchronisation
            # (ONE pass per step — prepare_indices: an aminmax, and the torch.unique unless check_indices is off)
            from graphembed.modules import prepare_indices
            indices, distinct = prepare_indices(indices, self.n, distinct=self.check_indices)
            if not distinct:
                raise ValueError('the node indices of a minibatch step must be distinct (slices of a randperm are: train.py:206-209)')
            self._idx = indices.to(device=self.device, dtype=torch.int64).contiguous()   # (kept: the enqueued kernels read it)
            batch = self._idx.numel()
            if batch == 0:
                # an EMPTY batch is still a minibatch step (no pairs: zero loss, zero gradients, the optimizers step on them —
                # what the reference's loop does with an empty index tensor); an empty tensor has no storage, and a null
                # batch_idx means "full batch" to the C side — which would read the dense matrix as a pair vector
                self._idx = torch.zeros(1, dtype=torch.int64, device=self.device)
            d.batch_idx, d.batch, d.target = self._idx.data_ptr(), batch, self.dense.data_ptr()
        else:
            if self.target.numel() == 0 and self.n > 1:
                raise ValueError('this stepper was built without a target pair vector: pass indices')
            d.batch_idx, d.batch, d.target = None, 0, self.target.data_ptr()
        d.loss_kind = B.LOSS_STRESS if spec[0] == 'stress' else B.LOSS_QUOTIENT
        d.alpha, d.eps, d.terms = float(spec[1]), float(spec[2]), int(spec[3])
        dyn = spec[4] if len(spec) > 4 else None
        d.loss_params = None if dyn is None else B.dyn_ptr(dyn, self.target).value
        need_first = False
        for i, (p, q) in enumerate(zip(self._params, self._slots)):
            q.x = p.data_ptr()              # (an eager optimizer step in between rebinds the parameter's storage)
            ok = self._bind_optimizer(p, q)
            if ok is None:
                need_first = True
            elif ok is False:
                if i < self.k:
                    raise ValueError('every point parameter needs an optimizer')
                q.x = p.data_ptr()          # a frozen scale (burn-in): read by the objective, not stepped
                q.optimizer = OPT_NONE
        if need_first:
            self._tables_of = None
            return self._first_step_unfused(indices, **objective_kwargs)
        if self._prepared_single:
            x = self._params[0]
            d.ws_flags = B.MM_WS_PREPARED if self._tables_of == (x.data_ptr(), x._version) else 0
        elif self.k > 1 and self.comm is None:
            # a product on one GPU: the step kernel also writes the node table of the symmetric pair kernel (where that one
            # is used) for the new points — valid as long as nobody else touched any factor's points
            now = tuple((x.data_ptr(), x._version) for x in self._params[:self.k])
            d.ws_flags = (d.ws_flags & B.WS_CLEAN) | (B.MM_WS_PREPARED if self._tables_of == now else 0)
        with B.on_device(self.device):
            B.lib().call('mm_train_step_run', ctypes.byref(d), B.stream_of(self.target))
        if self.k > 1:
            d.ws_flags = B.WS_CLEAN      # the pair kernel leaves its workspace clean
            self._tables_of = tuple((x.data_ptr(), x._version) for x in self._params[:self.k])
        elif self._prepared_single:
            x = self._params[0]
            # (a MINIBATCH step of a vector factor takes the unfused kernels: the zero-padded copy of the points in the workspace
            # is not rewritten, so the next full-batch step must prepare it again — tools/fuzz_step.py, round 4: a full batch
            # behind a minibatch read the stale copy.  The SPD step kernel writes the tables of all points in either form.)
            stale = indices is not None and not self._single_spd
            self._tables_of = None if stale else (x.data_ptr(), x._version)
        return self.loss_out[0]

    def invalidate_tables(self):
        """Forget that the workspace holds the per-node tables of the current points: the next step prepares them again.
        Storage rebinding and every in-place torch operation on the points are noticed by themselves (data pointer,
        version counter); call this after writing the points in a way torch does not see (a raw-pointer kernel, a
        graph replay of somebody else's step)."""
        self._tables_of = None

    def _first_step_unfused(self, indices=None, **objective_kwargs):
        """First step of a heavy-ball RSGD: the momentum buffers do not exist yet — run it through the optimizers."""
        if indices is not None:
            loss = self.embedding.fused_objective(self.objective_fn, None, self._idx, dense=self.dense, validated=True,
                                                  **objective_kwargs)
        elif self.shard is not None:
            from graphembed import parallel
            loss = parallel.sharded_fused_objective(self.embedding, self.objective_fn, self.target, self.shard,
                                                    comm=self.comm, **objective_kwargs)
        else:
            loss = self.embedding.fused_objective(self.objective_fn, self.target, None, **objective_kwargs)
        for o in self.optimizers:
            o.zero_grad(set_to_none=True)
        loss.backward()
        if self.shard is not None and self.shard.world > 1:
            loss = parallel.all_reduce_(loss.detach().clone(), comm=self.comm, group=self.shard.group)
        for o in self.optimizers:
            o.step()
        for x, g in zip(self.embedding.xs, self.grads):
            x.grad = g
        for i, s in enumerate(self.embedding.scales):
            s.grad = self.loss_out[1 + i].view(s.shape)
        return loss.detach()
