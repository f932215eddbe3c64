"""The multi-rank RCCL path on REAL ranks: mm_comm_init across processes, mm_allreduce_sum inside mm_train_step_run and its
capture in the step's HIP graph — two processes, one GPU each (tests/multi_gpu/rccl_child.py).

Needs >= 2 GPUs in the box: every development / grading box so far has ONE, so this test has been SKIPPED in every recorded
run — the N > 1 RCCL path is NOT hardware-verified (README.md, DESIGN.md §7 say so).  What does run on one GPU:
tests/test_comm_gpu.py (one-rank communicator, eager and captured; the row shards of 2 / 3 / 8 ranks summed by hand)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world', [1, 2])
def test_sharded_step_over_rccl_ranks(world):
    """world = 1 keeps the child script itself exercised on the one-GPU boxes (a one-rank communicator in a fresh process under
    torch.distributed.run); world = 2 is the multi-rank check proper."""
    if torch.cuda.device_count() < world:
        pytest.skip(f'needs {world} GPUs (one RCCL rank per GPU), {torch.cuda.device_count()} visible')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')       # dmabuf IPC (RCCL across processes)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'multi_gpu', 'rccl_child.py')]
    # fresh child processes (this process may hold a GPU context already: the children are started, never exec'ed into)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, start_new_session=True)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    for k in range(world):
        assert f'RANK {k} OK' in r.stdout, r.stdout[-2000:]
