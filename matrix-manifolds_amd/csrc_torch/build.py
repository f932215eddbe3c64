"""Build lib/_mm_autograd.so — the C++ autograd nodes of csrc_torch/mm_autograd.cpp (host code only: g++ against the torch
headers of this interpreter; the HIP library is bound at run time).  `python build.py` or build_if_stale() from __graft_entry__."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'mm_autograd.cpp')
OUT = os.path.join(os.path.dirname(HERE), 'lib', '_mm_autograd.so')
STAMP = os.path.join(os.path.dirname(HERE), 'lib', '_mm_autograd.stamp')   # torch version + source hash the binary was built from


def _stamp_of():
    """What the binary depends on besides its source: the torch it was compiled against (headers and libraries) — a binary
    built for another torch loads and then calls through mismatched types."""
    import hashlib
    import torch
    return f'torch {torch.__version__}\nsource {hashlib.sha256(open(SRC, "rb").read()).hexdigest()}\n'


def is_current():
    try:
        return os.path.isfile(OUT) and open(STAMP).read() == _stamp_of()
    except OSError:
        return False


def build_if_stale(force=False):
    if not force and is_current():
        return OUT
    import pybind11
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    cmd = ['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
           '-DTORCH_EXTENSION_NAME=_mm_autograd', '-DTORCH_API_INCLUDE_EXTENSION_H',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}']
    cmd += ['-I' + p for p in ce.include_paths()] + ['-I' + os.path.join(rocm, 'include'), '-I' + sysconfig.get_paths()['include'],
                                                    '-I' + pybind11.get_include()]
    cmd += [SRC, '-o', OUT, '-L' + tlib, '-ltorch', '-ltorch_cpu', '-ltorch_python', '-lc10', '-lc10_hip', '-ltorch_hip', '-ldl',
            '-Wl,-rpath,' + tlib]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if os.path.isfile(STAMP):
        os.remove(STAMP)
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as f:
        f.write(_stamp_of())
    return OUT


if __name__ == '__main__':
    print(build_if_stale(force='--force' in sys.argv))
