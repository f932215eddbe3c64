"""The drop-in boundary is a C ABI: a plain-C client (gcc, no torch, no C++) compiles against include/mm_manifolds.h,
links libmm_manifolds.so and — on a GPU box — reproduces closed-form SPD distances and gradients through it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, 'matrix-manifolds_amd', 'lib')
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')


def _build(tmp_path):
    if shutil.which('gcc') is None or not os.path.isdir(os.path.join(ROCM, 'include', 'hip')):
        pytest.skip('gcc / ROCm headers not available')
    exe = str(tmp_path / 'c_abi_smoke')
    cmd = ['gcc', '-std=c11', '-O1', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I' + os.path.join(ROOT, 'include'),
           '-I' + os.path.join(ROCM, 'include'), os.path.join(ROOT, 'tests', 'c_abi', 'smoke.c'), '-o', exe,
           '-L' + LIBDIR, '-lmm_manifolds', '-L' + os.path.join(ROCM, 'lib'), '-lamdhip64', '-lm',
           '-Wl,-rpath,' + LIBDIR, '-Wl,-rpath,' + os.path.join(ROCM, 'lib')]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_c_client_compiles_and_links(tmp_path):
    """The header is valid C11 and every symbol the client uses resolves against the shared library."""
    _build(tmp_path)


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    exe = _build(tmp_path)
    # The client maps /opt/rocm's libamdhip64 and an RCCL of its own — hundreds of megabytes that a fresh box pages in
    # from cold storage (the first `import torch` of a box takes minutes for the same reason): a generous limit, and the
    # RCCL copy PyTorch has already pulled into the page cache instead of a second one (MM_RCCL_LIB, csrc/comm.hip).
    env = dict(os.environ)
    try:
        import torch
        rccl = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
        if os.path.isfile(rccl):
            env.setdefault('MM_RCCL_LIB', rccl)
    except ImportError:
        pass
    res = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    assert 'C ABI smoke: ok' in res.stdout
