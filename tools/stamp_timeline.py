#!/usr/bin/env python3
"""Diagnostic: timeline of the SPD backward kernel's workgroups from in-kernel clock stamps (build
`tools/snap_make.sh stamp -DMM_BWD_STAMP`, run with MM_MANIFOLDS_LIB=.../libmm_stamp.so).
Prints residency over time, workgroup durations, the per-CU load and the SHADER CLOCK (s_memtime cycles per microsecond
of s_memrealtime).   python tools/stamp_timeline.py [n] [--warm] [--d=D] [--shard=R/N]
--warm: the stamped launch is the last of >= 60 ms of back-to-back graph replays — the clock regime of bench.py's timed
region (Clocks); without it: the 7th launch after seconds of host-side input generation (cold clocks)."""
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


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n = int(args[0]) if args else 5000
    opt = dict(a[2:].split('=') for a in sys.argv[1:] if a.startswith('--') and '=' in a)
    d = int(opt.get('d', 3))
    rank, world = (int(v) for v in opt.get('shard', '0/1').split('/'))
    dev = torch.device('cuda', 0)
    wl = bench.PdistWorkload(d, n, torch.float32, 0.1, world, rank, dev)
    print(f'SPD({d}) n = {n}, rank {rank} of {world}: rows {wl.rows}, {wl.hi - wl.lo} pairs')
    if '--warm' in sys.argv:
        import time
        graph, _ = bench.graph_of(wl.kernels, bench.Fence(1))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.06:
            for _ in range(16):
                graph.replay()
            torch.cuda.synchronize()
        for _ in range(64):       # the stamps read below are those of the last launch of an uninterrupted burst
            graph.replay()
        torch.cuda.synchronize()
        print('regime: warm (last launch of a 64-replay burst after 60 ms of replays)')
    else:
        for _ in range(5):
            wl.kernels()
        torch.cuda.synchronize()
        wl.kernels()
        torch.cuda.synchronize()
        print('regime: cold (7th eager launch after input generation)')
    raw = B.lib()._lib
    buf = np.zeros(4 * 16384, dtype=np.uint64)
    fn = raw.mm_dbg_read_bwd_stamps
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    st = buf.reshape(-1, 4)
    st = st[st[:, 1] > 0]
    t0, t1 = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64)
    base = t0.min()
    t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0      # s_memrealtime ticks at 100 MHz -> us
    hw = (st[:, 2] >> np.uint64(32)).astype(np.int64)
    xcc = (st[:, 2] & np.uint64(0xffffffff)).astype(np.int64) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    dur = t1 - t0
    cyc = st[:, 3].astype(np.int64)
    print(f'{len(st)} workgroups; kernel span {t1.max():.1f} us; workgroup duration us: min {dur.min():.1f} median {np.median(dur):.1f} '
          f'max {dur.max():.1f}; shader clock {np.median(cyc / np.maximum(dur, 1e-3)) / 1e3:.2f} GHz (median cycles / us)')
    print('start times us: percentiles 10/50/90/99/100:', np.percentile(t0, [10, 50, 90, 99, 100]).round(1))
    edges = np.linspace(0, t1.max(), 41)
    res = [(np.minimum(t1, b) - np.maximum(t0, a)).clip(min=0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    print('resident workgroups per 1/40 of the span:', ' '.join(f'{r:.0f}' for r in res))
    ncu = len(np.unique(cu))
    per = np.bincount(np.unique(cu, return_inverse=True)[1])
    print(f'{ncu} distinct CUs; workgroups per CU: min {per.min()} median {np.median(per):.0f} max {per.max()}')
    busy = np.zeros(ncu)
    inv = np.unique(cu, return_inverse=True)[1]
    np.add.at(busy, inv, dur)
    print(f'sum of workgroup durations per CU us: min {busy.min():.0f} median {np.median(busy):.0f} max {busy.max():.0f}')
    # concurrency per CU at the median time
    tm = 0.5 * t1.max()
    live = (t0 <= tm) & (t1 > tm)
    print(f'at t = {tm:.1f} us: {live.sum()} workgroups live, per CU max {np.bincount(inv[live], minlength=ncu).max()} '
          f'median {np.median(np.bincount(inv[live], minlength=ncu)):.0f}')
    first = t0 < 2.0
    print(f'workgroups started in the first 2 us: {first.sum()}; their durations us: min {dur[first].min():.1f} median {np.median(dur[first]):.1f} max {dur[first].max():.1f}')
    if hasattr(raw, 'mm_dbg_read_bwd_marks'):
        mk = np.zeros(3 * 16384, dtype=np.uint64)
        fm = raw.mm_dbg_read_bwd_marks
        fm.restype, fm.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
        if fm(mk.ctypes.data, mk.nbytes) == 0:
            mk = mk.reshape(-1, 3)[:len(buf.reshape(-1, 4))][buf.reshape(-1, 4)[:, 1] > 0].astype(np.float64)
            ghz = np.median(cyc / np.maximum(dur, 1e-3)) / 1e3
            ph = np.stack([mk[:, 0], mk[:, 1] - mk[:, 0], mk[:, 2] - mk[:, 1], cyc - mk[:, 2]], axis=1) / (ghz * 1e3)
            names = ['cut + find', 'operands until first row', 'rows', 'after last row (flush, barrier)']
            print('phases of wavefront 0, us (median / p90 / max): ' + '; '.join(
                f'{nm} {np.median(ph[:, k]):.2f} / {np.percentile(ph[:, k], 90):.2f} / {ph[:, k].max():.2f}' for k, nm in enumerate(names)))
    if 'dump' in opt:   # raw per-workgroup records (row = blockIdx.x): start us, end us, CU id, XCC, cycles
        np.save(opt['dump'], np.stack([t0, t1, cu.astype(np.float64), xcc.astype(np.float64), cyc.astype(np.float64)], axis=1))
    order = np.argsort(t0)
    print('first 12 starts:', t0[order][:12].round(2), ' last 12 ends:', np.sort(t1)[-12:].round(1))


if __name__ == '__main__':
    main()
