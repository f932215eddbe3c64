"""The collective of the sharded training path: a process-lifetime RCCL communicator behind the C ABI
(`mm_comm_init` / `mm_allreduce_sum`, csrc/comm.hip) — one process per GPU, one all-reduce(sum) of
{point gradients, loss, scale gradients} per step over xGMI.

It replaces the reference's only parallel call site, `torch.nn.DataParallel` around `BatchedObjective`
(graphembed/graphembed/train.py:107-109).  `torch.distributed` is used for the RENDEZVOUS only (handing rank 0's
128-byte token to the other ranks through whatever process group exists — gloo or nccl); the data path is the
library's own communicator, so the collective can sit inside `mm_train_step_run` and inside a captured HIP graph:

    comm = Communicator.from_torch_distributed(device)       # collective: every rank calls it
    comm.all_reduce_(flat)                                    # in place, on the current stream, no host sync

A communicator of world size 1 is valid (single-GPU runs exercise the same code path).  RCCL needs one GPU per
rank: ranks that share a device (the gloo dry runs of the N > 1 path on a one-GPU box) cannot build one — use
`torch.distributed` there (`graphembed.parallel` falls back to it when no communicator is installed).
"""
import ctypes

import torch

from graphembed import _backend as B

ID_BYTES = 128


class Communicator:

    def __init__(self, rank, world, unique_id, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise B.BackendError('the RCCL communicator lives on an MI355X: pass a cuda device')
        index = device.index if device.index is not None else torch.cuda.current_device()
        if len(unique_id) != ID_BYTES:
            raise ValueError(f'unique_id must be the {ID_BYTES}-byte token of Communicator.unique_id()')
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device('cuda', index)
        self._handle = ctypes.c_void_p()
        token = (ctypes.c_char * ID_BYTES).from_buffer_copy(bytes(unique_id))
        with B.on_device(self.device):
            _call('mm_comm_init', ctypes.byref(self._handle), self.rank, self.world, token, index)
            # first use outside any graph capture: RCCL sets up its channels lazily
            probe = torch.zeros(8, dtype=torch.float32, device=self.device)
            self.all_reduce_(probe)

    @staticmethod
    def unique_id():
        """The rendezvous token (rank 0 creates it; every rank passes the same one to the constructor)."""
        buf = (ctypes.c_char * ID_BYTES)()
        _call('mm_comm_unique_id', buf)
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, device, group=None):
        """Collective over `group` (default: the world): rank 0's token travels through torch.distributed."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return cls(0, 1, cls.unique_id(), device)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group,
                                   device=torch.device(device) if dist.get_backend(group) == 'nccl' else None)
        return cls(rank, world, box[0], device)

    @property
    def handle(self):
        """The `mm_comm_t` for `mm_train_step.comm` / `mm_allreduce_sum`."""
        if not self._handle:
            raise B.BackendError('communicator already destroyed')
        return self._handle

    def all_reduce_(self, t):
        """In-place sum over the ranks, enqueued on the current stream of `t`'s device (fp32 / fp64, contiguous)."""
        B.require_gpu(t)
        if not t.is_contiguous():
            raise ValueError('all_reduce_ needs a contiguous tensor (it is reduced in place)')
        if t.device != self.device:
            raise B.BackendError(f'tensor on {t.device}, communicator on {self.device}')
        with B.on_device(t.device):
            _call('mm_allreduce_sum', self.handle, B.dtype_code(t), B.ptr(t), t.numel(), B.stream_of(t))
        return t

    def destroy(self):
        if self._handle:
            h, self._handle = self._handle, ctypes.c_void_p()
            _call('mm_comm_destroy', h)

    def __del__(self):
        try:
            self.destroy()
        except Exception:  # noqa: BLE001 — interpreter shutdown
            pass


def _call(name, *args):
    lib = B.lib()
    rc = lib.raw(name)(*args)
    if rc == -3:
        raise B.BackendError(f'{name} failed: {lib.raw("mm_comm_last_error")().decode(errors="replace")}')
    if rc != 0:
        kind = {-1: 'invalid argument', -2: 'unsupported size/dtype'}.get(rc, f'hipError_t {rc}')
        raise B.BackendError(f'{name} failed: {kind}')


def available():
    """True if the library could bind RCCL (librccl on the loader path or already mapped by PyTorch)."""
    return bool(B.lib().raw('mm_comm_available')())
