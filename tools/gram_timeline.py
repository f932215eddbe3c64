#!/usr/bin/env python3
"""Diagnostic: where a wavefront of the symmetric matrix-core backward (vec_gram_bwd_sym_f32_kernel) spends its cycles.
Build vec_gram.hip with -DMM_GRAM_STAMP into a variant library and run with MM_MANIFOLDS_LIB pointing at it:
    python3 tools/gram_timeline.py [n] [m]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from graphembed import _backend as B  # noqa: E402
from graphembed import manifolds as M  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4039
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    man = M.Lorentz(m)
    torch.manual_seed(0)
    x = man.rand(n, out=torch.empty(0, device='cuda')).requires_grad_()
    g = torch.randn(n * (n - 1) // 2, device='cuda')
    for _ in range(6):
        d2 = man.pdist(x, squared=True)
        torch.autograd.grad(d2, x, g)
    torch.cuda.synchronize()
    raw = B.lib()._lib
    if len(sys.argv) > 3 and sys.argv[3] == 'fwd':
        fb = np.zeros(2048 * 4 * 16, dtype=np.uint64)
        ff = raw.mm_dbg_read_gramf_stamps
        ff.restype, ff.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
        assert ff(fb.ctypes.data, fb.nbytes) == 0
        fs = fb.reshape(-1, 16).astype(np.int64)
        fs = fs[fs[:, 14] > 0]
        life = fs[:, 14] - fs[:, 0]
        print(f'forward: {len(fs)} live wavefronts; life cycles min {life.min()} median {int(np.median(life))} max {life.max()}')
        print(f'  A operand + first B request: median {int(np.median(fs[:, 1] - fs[:, 0]))}')
        for t in range(4):
            ok = fs[:, 4 + 3 * t] > 0
            if not ok.any():
                continue
            prev = fs[ok, 1] if t == 0 else fs[ok, 1 + 3 * t]
            print(f'  tile {t}: {ok.sum()} wavefronts; B wait {int(np.median(fs[ok, 2 + 3 * t] - prev))}; MFMAs {int(np.median(fs[ok, 3 + 3 * t] - fs[ok, 2 + 3 * t]))}; '
                  f'acosh + stores issued {int(np.median(fs[ok, 4 + 3 * t] - fs[ok, 3 + 3 * t]))}')
        last = np.max(fs[:, 2:14], axis=1)
        print(f'  stores drained (vmcnt(0)) after the last tile: median {int(np.median(fs[:, 14] - last))}')
        end = fs[:, 15]
        print(f'  wavefront END spread by the realtime clock: {(end.max() - end.min()) / 100.0:.1f} us')
        return
    buf = np.zeros(1024 * 4 * 26, dtype=np.uint64)
    fn = raw.mm_dbg_read_gram_stamps
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    raw_st = buf.reshape(-1, 26)
    st = raw_st.astype(np.int64)
    keep = st[:, 22] > 0
    st, raw_st = st[keep], raw_st[keep]
    print(f'{len(st)} wavefronts with stamps ({len(st) // 4} workgroups)')
    total = st[:, 22] - st[:, 0]
    print(f'wavefront life, cycles: min {total.min()} median {int(np.median(total))} max {total.max()}')
    print(f'prologue (operands of the column block, zeroing, barrier): median {int(np.median(st[:, 1] - st[:, 0]))}')
    names = ['loads issued -> arrived', 'masks + Gram MFMA + dout/dq', 'ACC_J MFMAs issued + W to LDS', 'W X_J MFMAs + LDS accumulate',
             'barrier wait']
    for t in range(4):
        base = 2 + 5 * t
        live = st[:, base] > 0
        prev = st[live, base - 1]
        row = []
        for p in range(5):
            cur = st[live, base + p]
            ok = cur > 0
            row.append(int(np.median((cur - prev)[ok])) if ok.any() else -1)
            prev = np.where(ok, cur, prev)
        print(f'step {t}: {live.sum()} live wavefronts; median cycles per phase: ' + '; '.join(f'{nm} {v}' for nm, v in zip(names, row)))
    print(f'epilogue (flush of both accumulators): median {int(np.median(st[:, 22] - st[:, 21]))}')
    hw = (raw_st[:, 24] >> np.uint64(32)).astype(np.int64)
    xcc = (raw_st[:, 24] & np.uint64(0xffffffff)).astype(np.int64) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    simd = (hw >> 4) & 3
    ucu, inv = np.unique(cu, return_inverse=True)
    per = np.bincount(inv)
    print(f'{len(ucu)} distinct CUs host the {len(st)} wavefronts; wavefronts per CU: min {per.min()} median {int(np.median(per))} max {per.max()}; '
          f'histogram {dict(zip(*np.unique(per, return_counts=True)))}')
    for k in sorted(set(per)):
        sel = per[inv] == k
        print(f'  CUs with {k} wavefronts: wavefront life median {int(np.median(total[sel]))} max {total[sel].max()}')
    print('wavefront life percentiles 50/75/90/95/99/100:', np.percentile(total, [50, 75, 90, 95, 99, 100]).astype(int))
    for k in sorted(set(per)):
        sel = per[inv] == k
        sums = np.zeros(5)
        for t in range(4):
            base = 2 + 5 * t
            prev = st[sel, base - 1]
            for p5 in range(5):
                cur = st[sel, base + p5]
                ok = (cur > 0) & (st[sel, base] > 0)
                if ok.any():
                    sums[p5] += np.median((cur - prev)[ok])
                prev = np.where(ok, cur, prev)
        print(f'  CUs with {k} wavefronts, median cycles per phase summed over the 4 steps: ' + '; '.join(f'{nm} {int(v)}' for nm, v in zip(names, sums))
              + f'; prologue {int(np.median((st[sel, 1] - st[sel, 0])))}; epilogue {int(np.median((st[sel, 22] - st[sel, 21])))}')
    end = st[:, 23]
    print(f'kernel span by the realtime clock: {(end.max() - end.min()) / 100.0:.1f} us between the first and the last wavefront END')


if __name__ == '__main__':
    main()
