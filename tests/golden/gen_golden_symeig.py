"""Golden vectors of SymmetricPositiveDefinite.symeig from the REAL reference (development container only).
    PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0 python tests/golden/gen_golden_symeig.py
graphembed/manifolds/spd.py:35-41, 63-64 (linalg/fast.py:53-91 closed forms for n = 2, 3; LAPACK otherwise), n = 2 … 9,
fp32 + fp64, points at the reference's initialisation and well-conditioned random points; output tests/golden/symeig.npz."""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from graphembed.manifolds import SymmetricPositiveDefinite  # noqa: E402
from gen_golden import np_, DT  # noqa: E402


def main():
    out = {}
    for d in range(2, 10):
        for dname in DT:
            torch.set_default_dtype(DT[dname])
            torch.manual_seed(zlib.crc32(repr((d, dname, 'symeig')).encode()) % (2**31))
            man = SymmetricPositiveDefinite(d)
            for init in ('rand', 'wide'):
                n = 70
                if init == 'rand':
                    x = man.rand(n)
                else:
                    a = torch.rand(n, d, d)
                    x = a @ a.transpose(1, 2) + torch.eye(d)
                w = man.symeig(x.detach())
                w = w[0] if isinstance(w, (tuple, list)) else w
                assert torch.isfinite(w).all()
                out[f'spd{d}/{dname}/{init}/x'] = np_(x)
                out[f'spd{d}/{dname}/{init}/w'] = np_(torch.sort(w, dim=-1).values)   # (the closed forms do not promise an order)
    np.savez_compressed(os.path.join(HERE, 'symeig.npz'), **out)
    print(len(out), 'arrays')
    torch.set_default_dtype(torch.float32)


if __name__ == '__main__':
    main()
