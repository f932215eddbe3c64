#!/usr/bin/env python3
"""One workload of bench.py, alone in a process, for rocprofv3 (kernel names do not carry n):
    python3 tools/profile_case.py pdist D N f32|f64 IR [steps]      # SPD(D).pdist fwd + bwd, ||log X|| = IR
    python3 tools/profile_case.py loss  D N f32|f64 [steps]         # fused QuotientLoss step (BASELINE config 5)
    python3 tools/profile_case.py vec   M N f32|f64 KIND [steps]    # KIND = lorentz | sphere | euclidean: pdist fwd + bwd
    python3 tools/profile_case.py step  D N f32|f64 [steps]         # full SPD(D) training step through mm_train_step_run (StressLoss + RSGD)
    python3 tools/profile_case.py product N f32|f64 [steps]         # BASELINE config 4: H^5 x S^5 x SPD(2) training step, mixed-manifold pair kernel
    python3 tools/profile_case.py vstep M N f32|f64 KIND [steps]    # full training step of one vector factor (mm_train_step_run)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402


def vec_case(m, n, dt, which, steps, dev):
    """facebook-class graph, BASELINE config 2 (n = 4039, Lorentz(11)): squared distances and their backward."""
    from graphembed import manifolds as M
    man = {'lorentz': M.Lorentz, 'sphere': M.Sphere, 'euclidean': M.Euclidean}[which](m)
    torch.manual_seed(0)
    x = man.rand(n, out=torch.empty(0, device=dev, dtype=dt)).requires_grad_()
    g = torch.randn(n * (n - 1) // 2, device=dev, dtype=dt)
    def step():
        d2 = man.pdist(x, squared=True)
        torch.autograd.grad(d2, x, g)
    run_warm(step, steps)


def main():
    dev = torch.device('cuda', 0)
    if sys.argv[1] == 'product':
        from graphembed import manifolds as M
        n, dt = int(sys.argv[2]), {'f32': torch.float32, 'f64': torch.float64}[sys.argv[3]]
        wl = bench.TrainStepWorkload([M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], n, dt, dev)
        return run_warm(wl.kernels, int(sys.argv[4]) if len(sys.argv) > 4 else 10)
    kind, d, n, dt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), {'f32': torch.float32, 'f64': torch.float64}[sys.argv[4]]
    if kind == 'step':
        from graphembed import manifolds as M
        wl = bench.TrainStepWorkload([M.SymmetricPositiveDefinite(d)], n, dt, dev)
        return run_warm(wl.kernels, int(sys.argv[5]) if len(sys.argv) > 5 else 10)
    if kind == 'vstep':
        from graphembed import manifolds as M
        man = {'lorentz': M.Lorentz, 'sphere': M.Sphere, 'euclidean': M.Euclidean}[sys.argv[5]](d)
        wl = bench.TrainStepWorkload([man], n, dt, dev)
        return run_warm(wl.kernels, int(sys.argv[6]) if len(sys.argv) > 6 else 10)
    if kind == 'vec':
        return vec_case(d, n, dt, sys.argv[5], int(sys.argv[6]) if len(sys.argv) > 6 else 10, dev)
    if kind == 'pdist':
        wl = bench.PdistWorkload(d, n, dt, float(sys.argv[5]), 1, 0, dev)
        steps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    else:
        wl = bench.FusedLossWorkload(d, n, dt, 1, 0, dev)
        steps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    run_warm(wl.kernels, steps)


def run_warm(fn, steps):
    """`steps` launches of the step — replayed from a captured graph, and for at least 60 ms, when more than a handful are
    asked for: the averages rocprofv3 reports are then those of a busy device (issued one by one through Python autograd
    the device idles half of the time, and a few dozen launches after seconds of host-side input generation run at lower
    clocks — DESIGN.md §4).  Counter-collection passes ask for 3 steps and get 3 eager ones."""
    import time
    if steps <= 5:
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return
    graph, _ = bench.graph_of(fn, bench.Fence(1))
    t0 = time.perf_counter()
    done = 0
    while done < steps or time.perf_counter() - t0 < 0.06:
        graph.replay()
        done += 1
        if done % 16 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
