"""One rank of the multi-GPU check of the RCCL path (tests/test_comm_multi_gpu.py starts WORLD of these through
torch.distributed.run, one per GPU, in fresh processes).  Each rank:

  1. builds the library's communicator (mm_comm_init) with the token handed over through torch.distributed (gloo: the data
     path never touches torch.distributed),
  2. all-reduces a known vector through mm_allreduce_sum and checks the sum,
  3. runs the sharded one-call training step (mm_train_step_run with this rank's rows and the communicator) eagerly and then
     captured in ONE HIP graph (kernels + all-reduce + optimizer), SPD(3) / Lorentz(11) / SPD(4), fp32 and fp64,
  4. runs the SAME steps unsharded on its own GPU (the single-GPU reference) and compares losses and parameters,
  5. compares its parameters with rank 0's (replicas must stay identical: every rank applies the same update to the same
     all-reduced gradient — train.py:107-109 needs a broadcast per step for that).

Prints `RANK r OK` and exits 0, or raises."""
import copy
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def embedding(case, n, dt):
    from graphembed import manifolds as M
    from graphembed.modules import ManifoldEmbedding
    mk = {'spd3': lambda: [M.SymmetricPositiveDefinite(3)], 'spd4': lambda: [M.SymmetricPositiveDefinite(4)],
          'lorentz11': lambda: [M.Lorentz(11)]}[case]
    torch.set_default_dtype(dt)
    try:
        torch.manual_seed(11)                      # the same embedding and targets on every rank
        with torch.device('cuda'):
            emb = ManifoldEmbedding(n, mk())
            with torch.no_grad():
                emb.perturb(0.3)
            target = torch.rand(n * (n - 1) // 2) * 0.9 + 0.05
    finally:
        torch.set_default_dtype(torch.float32)
    return emb, target


def optimizers(emb, adam):
    from graphembed.optim import RiemannianAdam, RiemannianSGD
    if adam:
        return [RiemannianAdam(list(emb.xs), lr=1e-2, exact=True, max_grad_norm=20), RiemannianAdam(list(emb.scales), lr=1e-3, max_grad_norm=500)]
    return [RiemannianSGD(list(emb.xs), lr=0.01, exact=True, max_grad_norm=20), RiemannianSGD(list(emb.scales), lr=1e-4, max_grad_norm=500)]


def main():
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    dist.init_process_group('gloo')              # rendezvous only
    from graphembed.comm import Communicator
    from graphembed.native_step import NativeTrainStep
    from graphembed.objectives import QuotientLoss, StressLoss
    from graphembed.parallel import PairShard
    comm = Communicator.from_torch_distributed(dev)
    assert (comm.rank, comm.world) == (rank, world)

    # 2. the collective itself
    for dt in (torch.float32, torch.float64):
        v = torch.arange(100003, dtype=dt, device=dev) * (rank + 1)
        comm.all_reduce_(v)
        torch.cuda.synchronize()
        want = torch.arange(100003, dtype=dt, device=dev) * (world * (world + 1) // 2)
        assert torch.equal(v, want), 'mm_allreduce_sum'

    # 3.-5. the sharded step against the unsharded one
    n = 190
    for case, adam in (('spd3', False), ('lorentz11', True), ('spd4', True)):
        for dt in (torch.float32, torch.float64):
            emb_ref, target = embedding(case, n, dt)
            emb_sh = copy.deepcopy(emb_ref)
            fn_ref, fn_sh = (QuotientLoss(), QuotientLoss()) if adam else (StressLoss(), StressLoss())
            if adam:
                fn_ref.on_device(dev), fn_sh.on_device(dev)
            plain = NativeTrainStep(emb_ref, fn_ref, target, optimizers(emb_ref, adam))
            shard = PairShard(n, world=world, rank=rank)
            sharded = NativeTrainStep(emb_sh, fn_sh, target, optimizers(emb_sh, adam), shard=shard, comm=comm)
            kw = dict(epoch=2, alpha=1.0)
            la, lb = [], []
            for _ in range(2):                       # eager
                la.append(float(plain(**kw)))
                lb.append(float(sharded(**kw)))
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()           # captured: kernels + all-reduce + optimizer in one graph
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                loss_g = sharded(**kw)
            for _ in range(3):
                la.append(float(plain(**kw)))
                graph.replay()
                lb.append(float(loss_g))
            eps = 1.2e-7 if dt == torch.float32 else 2.3e-16
            np.testing.assert_allclose(lb, la, rtol=2048 * eps, err_msg=f'{case} {dt}: losses of the sharded step')
            for a, b in zip(list(emb_ref.xs) + list(emb_ref.scales), list(emb_sh.xs) + list(emb_sh.scales)):
                a, b = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
                assert np.abs(a - b).max() <= 4096 * eps * max(np.abs(a).max(), 1.0), (case, dt, np.abs(a - b).max())
            # replicas: bit-identical parameters on every rank (same all-reduced gradient, same update)
            for p in list(emb_sh.xs) + list(emb_sh.scales):
                mine = p.detach().cpu()
                box = [mine if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                assert torch.equal(mine, box[0]), (case, dt, 'replicas diverged')
    comm.destroy()
    dist.barrier()
    dist.destroy_process_group()
    print(f'RANK {rank} OK', flush=True)


if __name__ == '__main__':
    main()
