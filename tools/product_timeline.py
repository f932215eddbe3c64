#!/usr/bin/env python3
"""Diagnostic: where a workgroup of the ordered mixed-manifold pair kernel (product_pair_kernel, BASELINE config 4) spends its
life.  Build product_pairs.hip with -DMM_PRODUCT_STAMP into a variant library and point MM_MANIFOLDS_LIB at it:
    MM_VARIANT_SRC=product_pairs.hip tools/snap_make.sh pstamp -DMM_PRODUCT_STAMP
    MM_MANIFOLDS_LIB=.../lib/variants/libmm_pstamp.so python3 tools/product_timeline.py [n]
Phases (shader cycles, lane 0 of wavefront 0 of every workgroup): prologue requests | staging + barrier | row loop | flushes |
loss partials; plus the spread of the workgroups' START times (100-MHz realtime clock) and the load per CU."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from graphembed import _backend as B  # noqa: E402
from graphembed import manifolds as M  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1025
    wl = bench.TrainStepWorkload([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], n, torch.float32, torch.device('cuda', 0))
    graph, _ = bench.graph_of(wl.kernels, bench.Fence(1))
    for _ in range(400):          # warm clocks, steady state
        graph.replay()
    torch.cuda.synchronize()
    raw = B.lib()._lib
    buf = np.zeros(8 * 4096, dtype=np.uint64)
    fn = raw.mm_dbg_read_product_stamps
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    st = buf.reshape(-1, 8)
    st = st[st[:, 5] > 0]
    c = st[:, :6].astype(np.int64)
    names = ['column data + row staging requested', 'rows staged in LDS, barrier passed', 'row loop', 'column flushes (LDS combine + atomics)', 'loss partials']
    life = c[:, 5] - c[:, 0]
    print(f'{len(st)} workgroups; life cycles: min {life.min()} median {int(np.median(life))} max {life.max()}')
    for k, nm in enumerate(names):
        d = c[:, k + 1] - c[:, k]
        print(f'  {nm:44s} median {int(np.median(d)):7d}  p10 {int(np.percentile(d, 10)):7d}  p90 {int(np.percentile(d, 90)):7d}')
    start = st[:, 6].astype(np.int64)
    t0 = start.min()
    rel = (start - t0) / 100.0
    print(f'workgroup START times (us after the first): median {np.median(rel):.2f}  p90 {np.percentile(rel, 90):.2f}  max {rel.max():.2f}')
    hw = st[:, 7]
    cu = ((hw >> 32) >> 8) & 0xF
    se = ((hw >> 32) >> 13) & 0x7
    xcc = hw & 0xF
    key = xcc.astype(np.int64) * 1000 + se.astype(np.int64) * 100 + cu.astype(np.int64)
    _, counts = np.unique(key, return_counts=True)
    print(f'compute units used: {len(counts)}; workgroups per CU: min {counts.min()} median {int(np.median(counts))} max {counts.max()}')
    # span of the launch by the realtime clock: first start to (last start + its life at ~2.4 GHz)
    end_est = rel + life / 2400.0
    print(f'estimated launch span: {end_est.max():.1f} us (last workgroup to finish started at {rel[np.argmax(end_est)]:.2f} us and lived {life[np.argmax(end_est)] / 2400.0:.2f} us)')


if __name__ == '__main__':
    main()
