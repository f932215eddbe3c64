"""Pair-range sharding across GPUs: one process per GPU, the embedding replicated,
the pair list [0,P) split into contiguous row ranges, ONE all-reduce(sum) per step.

This replaces the reference's only parallelism, `torch.nn.DataParallel` over
`BatchedObjective` (graphembed/graphembed/train.py:107-109), which per step does a
parameter broadcast, a scalar gather and a gradient reduce-add — and, because it
scatters *node* indices, silently drops every cross-chunk pair (train.py:203-213).
Here every pair is evaluated exactly once, each rank consumes only its slice of the
targets, and since all ranks apply the same deterministic optimizer step to the same
all-reduced gradient the replicas stay identical without any broadcast.

The collective is the library's own RCCL communicator (`graphembed.comm.Communicator`, the C ABI's
`mm_allreduce_sum` over xGMI) when one is passed or installed with `set_communicator`; without one it is
`torch.distributed` ("nccl" = RCCL; "gloo" runs the same code on CPUs — the tests — and for ranks that share
a GPU).  The message is n*point_size(+scales) elements (<= 1 MiB at the reference's sizes): latency-bound,
so it is issued once, on the compute stream, and it is capturable into the step's HIP graph.
"""
import torch
import torch.distributed as dist

from graphembed import _backend as B


_communicator = None


def set_communicator(comm):
    """Install the process-wide RCCL communicator (`graphembed.comm.Communicator`) used by `sync_grads` and
    the sharded objectives when none is passed explicitly; `None` goes back to torch.distributed."""
    global _communicator
    _communicator = comm


def get_communicator():
    return _communicator


def all_reduce_(t, comm=None, group=None):
    """In-place sum of `t` over the ranks: through the RCCL communicator if there is one, else torch.distributed."""
    comm = comm if comm is not None else _communicator
    if comm is not None:
        if comm.world > 1:
            comm.all_reduce_(t)
        return t
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return t


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


class PairShard:
    """The slice of the pair list owned by this rank for an n-point embedding."""

    def __init__(self, n, group=None, world=None, rank=None):
        if world is None:
            world, rank = world_info(group)
        self.n, self.world, self.rank, self.group = n, world, rank, group
        self.rows = B.shard_rows(n, world, rank)
        self.lo = B.pair_offset(n, self.rows[0])
        self.hi = B.pair_offset(n, self.rows[1])

    @property
    def num_pairs(self):
        return self.hi - self.lo

    def slice(self, pair_vector):
        """This rank's part of a full-length (P,) vector (targets, upstream gradients)."""
        return pair_vector[self.lo:self.hi]


class _SyncGrads(torch.autograd.Function):
    """Identity in the forward; in the backward all incoming gradients are packed into
    one flat buffer, summed over the ranks with a single all-reduce, and unpacked."""

    @staticmethod
    def forward(ctx, group, comm, *tensors):
        ctx.group, ctx.comm = group, comm
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        present = [g for g in grads if g is not None]
        comm = ctx.comm if ctx.comm is not None else _communicator
        if comm is not None:
            many = comm.world > 1
        else:
            many = dist.is_available() and dist.is_initialized() and dist.get_world_size(ctx.group) > 1
        if present and many:
            dtype = present[0].dtype
            for g in present[1:]:
                dtype = torch.promote_types(dtype, g.dtype)
            flat = torch.cat([g.reshape(-1).to(dtype) for g in present])
            all_reduce_(flat, comm=comm, group=ctx.group)  # the single collective of a step
            out, off = [], 0
            for g in grads:
                if g is None:
                    out.append(None)
                    continue
                k = g.numel()
                out.append(flat[off:off + k].view_as(g).to(g.dtype))
                off += k
            grads = tuple(out)
        return (None, None) + tuple(grads)


def sync_grads(*tensors, group=None, comm=None):
    """Return views of `tensors` whose gradients are all-reduced (sum) in one collective when backward
    reaches them — over the RCCL communicator `comm` (or the installed one), else over `group`."""
    return _SyncGrads.apply(group, comm, *tensors)


def sharded_compute_dists(embedding, shard, comm=None):
    """ManifoldEmbedding.compute_dists (modules.py:84-88) restricted to this rank's pair
    slice: sum_k softplus(s_k) * pdist_k(x_k, squared=True)[lo:hi].  Backward leaves the
    all-reduced (i.e. full) gradients in `x.grad` / `scale.grad` on every rank."""
    from torch.nn.functional import softplus
    params = list(embedding.xs) + list(embedding.scales)
    synced = sync_grads(*params, group=shard.group, comm=comm)
    k = len(embedding.xs)
    return sum(
        softplus(s) * man.pdist(x, squared=True, rows=shard.rows)
        for x, s, man in zip(synced[:k], synced[k:], embedding.manifolds))


def sharded_fused_objective(embedding, objective_fn, targets, shard, comm=None, **kwargs):
    """`objective_fn(targets, embedding.compute_dists())` of this rank's pair slice through the fused
    loss+gradient kernels (ManifoldEmbedding.fused_objective: one pass, no pair vector), or None when the
    configuration has no fused kernel.  `targets` is the full pair vector (sliced here) or already this
    rank's slice.  Backward leaves the all-reduced (i.e. full) gradients in `x.grad` / `scale.grad` on every
    rank — one collective; the returned loss is the LOCAL part (sum over ranks = the loss)."""
    if targets.numel() != shard.num_pairs:
        targets = shard.slice(targets)
    params = list(embedding.xs) + list(embedding.scales)
    synced = sync_grads(*params, group=shard.group, comm=comm)
    k = len(embedding.xs)
    return embedding.fused_objective(objective_fn, targets, None, rows=shard.rows,
                                     params=(list(synced[:k]), list(synced[k:])), **kwargs)
