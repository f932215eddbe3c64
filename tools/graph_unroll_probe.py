"""How much of a step is the boundary between two graph launches?  Captures U consecutive steps of a workload into ONE
hipGraph (U = 1, 2, 4, 8, 16) and reports the wall time per step of back-to-back replays, plus the host's enqueue time per
replay (no synchronisation inside the loop).
    python tools/graph_unroll_probe.py [headline|config4|config2|config3|all] [U ...]     # (U given: only those, e.g. under rocprofv3)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'matrix-manifolds_amd'))

import torch  # noqa: E402

import bench  # noqa: E402


US = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8, 16]


def probe(name, wl, total=960):
    fence = torch.cuda.synchronize
    out = []
    for u in US:
        def fn():
            r = None
            for _ in range(u):
                r = wl.kernels()
            return r
        graph, _ = bench.graph_of(fn, fence)
        reps = total // u
        for _ in range(max(3, 200 // u)):
            graph.replay()
        fence()
        best, host = 1e9, 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(reps):
                graph.replay()
            t1 = time.perf_counter()
            fence()
            t2 = time.perf_counter()
            best = min(best, (t2 - t0) / (reps * u))
            host = min(host, (t1 - t0) / reps)
        out.append(f'U={u}: {best * 1e6:.2f} us/step (host enqueue {host * 1e6:.1f} us/replay)')
        del graph
    print(name, '|', '; '.join(out), flush=True)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    dev = torch.device('cuda:0')
    from graphembed import manifolds as M
    if which in ('headline', 'all'):
        probe('headline SPD(3) n=5000 f32 pdist fwd+bwd', bench.PdistWorkload(3, 5000, torch.float32, 0.1, 1, 0, dev))
    if which in ('config4', 'all'):
        mans = [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)]
        probe('config 4 step n=1025 f32', bench.TrainStepWorkload(mans, 1025, torch.float32, dev))
    if which in ('config2', 'all'):
        probe('config 2 step Lorentz(11) n=4039 f32', bench.TrainStepWorkload([M.Lorentz(11)], 4039, torch.float32, dev))
    if which in ('config3', 'all'):
        probe('config 3 step SPD(3) n=5000 f32', bench.TrainStepWorkload([M.SymmetricPositiveDefinite(3)], 5000, torch.float32, dev))


if __name__ == '__main__':
    main()
