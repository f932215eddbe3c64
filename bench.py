#!/usr/bin/env python3
"""Headline benchmark: pairwise manifold-dist/sec (fwd+bwd), SPD(3), 5k-node graph.

One step = one pass of the hot path over all pairs of the synthetic embedding:
    d2 = SPD(3).pdist(x, squared=True);  d2.backward(g)
(`g` a fixed random upstream gradient, so no loss is fused in — SURVEY.md §8d),
with the pair list sharded by rows across the ranks and, for N > 1, ONE RCCL
all-reduce(sum) of the embedding gradient inside the timed region — issued through the
library's own communicator (mm_allreduce_sum, include/mm_manifolds.h) and captured in the
same hipGraph as the kernels.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: `python bench.py --gpus N` starts its N ranks itself (a child `torch.distributed.run`,
started before this process touches a GPU; the child's exit code is passed on, a run that
exceeds --launch-timeout is killed and exits 124).  Started BY `torch.distributed.run` (RANK in
the environment) it is one rank.  Rank 0 prints ONE JSON line.  Inputs are resident in HBM
before the timed region starts.  The reference gets its multi-GPU from one `python run.py`
(graphembed/graphembed/train.py:107-109, torch.nn.DataParallel).
"""
import argparse
import ctypes
import hashlib
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'matrix-manifolds_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

N_NODES = 5000
DIM = 3
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--n', type=int, default=N_NODES)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='no HIP-event bracketing of the kernels')
    ap.add_argument('--no-extra', action='store_true', help='headline only (no secondary workloads)')
    ap.add_argument('--graph', choices=['auto', 'on', 'off'], default='auto',
                    help='replay the step from a captured hipGraph (auto = on; falls back to eager launches '
                         'if capture is unavailable)')
    ap.add_argument('--launch-timeout', type=float, default=900.0,
                    help='N > 1 self-launch: seconds before the child job is killed (exit 124)')
    ap.add_argument('--rank-timeout', type=float, default=600.0,
                    help='a rank that is still running after this many seconds exits with code 3')
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torch.distributed.run around it
# ---------------------------------------------------------------------------------------------
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch(args, argv):
    """Spawn the N ranks as ONE child job and wait for it.  This process never initialises a GPU and
    never execs: it starts a child, waits (bounded) and exits with the child's code."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (RCCL / cross-process device memory)
    env['MM_BENCH_LAUNCHED'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)   # own process group: killable as a whole
    try:
        rc = proc.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        print(f'[bench] the {args.gpus}-rank job exceeded --launch-timeout {args.launch_timeout:.0f} s: killing it',
              file=sys.stderr, flush=True)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)   # exactly the group started above
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        rc = 124
    except KeyboardInterrupt:
        os.killpg(proc.pid, signal.SIGTERM)
        rc = 130
    sys.exit(rc)


# ---------------------------------------------------------------------------------------------
# synthetic workloads
# ---------------------------------------------------------------------------------------------
def synthetic_spd(n, d, seed, device, ir=0.1, dtype=None, with_g=True):
    """SPD.rand (spd.py:201-208): X = expm(U), ||vec U|| = ir (0.1 = the reference's init)."""
    import torch
    gen = torch.Generator().manual_seed(seed)
    m = d * (d + 1) // 2
    u = torch.randn(n, m, generator=gen, dtype=torch.float64)
    u = u / u.norm(dim=-1, keepdim=True) * ir
    iu = torch.triu_indices(d, d)
    U = torch.zeros(n, d, d, dtype=torch.float64)
    U[:, iu[0], iu[1]] = u / 2 ** 0.5
    U[:, iu[1], iu[0]] = u / 2 ** 0.5
    k = torch.arange(d)
    U[:, k, k] *= 2 ** 0.5
    X = torch.linalg.matrix_exp(U)
    dtype = dtype or torch.float32
    if not with_g:
        return X.to(dtype).to(device), None
    P = n * (n - 1) // 2
    g = torch.randn(P, generator=gen, dtype=torch.float32)
    return X.to(dtype).to(device), g.to(dtype).to(device)


def cpu_baseline(seed, n=N_NODES):
    """The oracle port (reference-faithful op sequence, oracle/ref_port.py) on this host's cores, on the
    metric's own workload (n = 5000, 12.5 M pairs).  The thread pool is chosen on a quarter-size probe
    (torch's intra-op pool does not scale to hundreds of threads on these element-wise ops), then the
    full size is timed twice with it and the faster pass is reported."""
    import torch
    from oracle import ref_port
    ncpu = os.cpu_count() or 1
    man = ref_port.SPD(DIM)

    def one(x, g):
        xr = x.clone().requires_grad_()
        t0 = time.perf_counter()
        d2 = man.pdist(xr, squared=True)
        d2.backward(g)
        return time.perf_counter() - t0

    xp, gp = synthetic_spd(n // 2, DIM, seed, 'cpu')
    best_threads, best = 1, float('inf')
    for threads in sorted({min(8, ncpu), min(32, ncpu), min(64, ncpu)}):
        torch.set_num_threads(threads)
        one(xp, gp)
        dt = one(xp, gp)
        if dt < best:
            best, best_threads = dt, threads
    torch.set_num_threads(best_threads)
    x, g = synthetic_spd(n, DIM, seed, 'cpu')
    P = n * (n - 1) // 2
    dt = min(one(x, g), one(x, g))
    return {'value': P / dt, 'unit': 'pairs/s', 'cores': best_threads, 'kind': 'port', 'host_cpus': ncpu,
            'seconds_per_pass': dt,
            'sample': f'SPD(3) fp32 reference-init, n={n} ({P} pairs), fwd+bwd, faster of 2 passes; thread pool '
                      f'(8/32/64) picked on an n={n // 2} probe; oracle/ref_port.py (torch CPU, reference op sequence)'}


def kernel_source_hash():
    """sha256 over the kernel sources: PMC figures are only quoted for the build they were measured on."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, 'matrix-manifolds_amd', 'csrc')
    for name in sorted(os.listdir(src)):
        if name.endswith(('.hip', '.hpp')):
            with open(os.path.join(src, name), 'rb') as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()[:16]


def stamped_pmc():
    """profiles/pmc_head.json (written by tools/pmc_stamp.py from rocprofv3 --pmc passes of this very
    command) if its source hash is that of the kernels in the tree, else None."""
    path = os.path.join(ROOT, 'profiles', 'pmc_head.json')
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    return rec if rec.get('kernel_source_hash') == kernel_source_hash() else None


# ---------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------
class Fence:
    def __init__(self, world):
        self.world = world

    def __call__(self):
        import torch
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()


def graph_of(fn, fence, warm=3, with_collective=False):
    """Capture fn() (a sequence of launches on the current stream) into a hipGraph; returns (graph, outputs).
    with_collective (any multi-rank run): the capture is thread-local — RCCL's proxy threads and the process group's
    watchdog thread may call the runtime while this thread records."""
    import torch
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    fence()
    graph = torch.cuda.CUDAGraph()
    kw = {'capture_error_mode': 'thread_local'} if with_collective else {}
    with torch.cuda.graph(graph, **kw):
        out = fn()
    fence()
    return graph, out


def all_agree(ok, world, dev):
    """True iff `ok` on every rank (a capture that fails on one rank must fail the mode for all of them)."""
    if world == 1:
        return ok
    import torch
    import torch.distributed as dist
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return flag.item() > 0


def collect_kernel_us(lib, names=(('fwd', 0), ('bwd', 1))):
    out = {}
    for name, kid in names:
        cnt, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
        lib.call('mm_prof_collect', kid, ctypes.byref(cnt), ctypes.byref(ms))
        out[name] = (ms.value / cnt.value * 1e3) if cnt.value else None
    return out


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


N_SIMD = 256 * 4             # MI355X: 256 CUs x 4 SIMDs
ISSUE_CYCLES = 2.0           # one instruction of a wavefront per ~2 cycles per SIMD, of any kind


def shader_clock_mhz(lib, run, dev, bursts=6):
    """Shader clock in the regime of the timed region: mm_prof_clock_probe (s_memtime cycles per 100-MHz s_memrealtime tick of
    a one-wavefront dependent chain) enqueued right behind a burst of replays of the timed step; median over a few bursts."""
    import torch
    from graphembed import _backend as B
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    vals = []
    for _ in range(bursts):
        for _ in range(16):
            run()
        lib.call('mm_prof_clock_probe', B.ptr(out), 20000, B.stream_of(out))
        torch.cuda.synchronize()
        c, r = (int(v) for v in out.tolist())
        if r > 0:
            vals.append(100.0 * c / r)
    return median(vals)


class PdistWorkload:
    """d2 = man.pdist(x, squared=True, rows=shard); grad = d d2 / d x . g  (+ one all-reduce for N > 1)."""

    def __init__(self, d, n, dtype, ir, world, rank, dev, seed=42, local_g=False, comm=None):
        import torch
        from graphembed import _backend as B
        from graphembed.manifolds import SymmetricPositiveDefinite
        self.d, self.n, self.world, self.rank, self.comm = d, n, world, rank, comm
        self.dtype = dtype
        self.rows = B.shard_rows(n, world, rank)
        self.lo, self.hi = B.pair_offset(n, self.rows[0]), B.pair_offset(n, self.rows[1])
        if local_g:   # (large n: only this rank's slice of the upstream gradient is ever generated)
            x, _ = synthetic_spd(n, d, seed, dev, ir=ir, dtype=dtype, with_g=False)
            gen = torch.Generator(device=dev).manual_seed(seed + 1 + rank)
            self.g_local = torch.randn(self.hi - self.lo, generator=gen, dtype=dtype, device=dev)
        else:
            x, g = synthetic_spd(n, d, seed, dev, ir=ir, dtype=dtype)   # replicated embedding, same on every rank
            self.g_local = g[self.lo:self.hi].contiguous()
            del g
        self.man = SymmetricPositiveDefinite(d)
        self.x = x.requires_grad_()
        self.P = n * (n - 1) // 2
        self.esz = torch.empty(0, dtype=dtype).element_size()

    def kernels(self):
        import torch
        d2 = self.man.pdist(self.x, squared=True, rows=self.rows)
        grad, = torch.autograd.grad(d2, self.x, self.g_local)
        return grad

    def reduce(self, t):
        """The single collective of a step: mm_allreduce_sum over the library's RCCL communicator, or
        torch.distributed when the ranks share a GPU (gloo dry run)."""
        from graphembed import parallel
        return parallel.all_reduce_(t, comm=self.comm)

    def eager_step(self):
        grad = self.kernels()
        if self.world > 1:
            self.reduce(grad)
        return grad


class FusedLossWorkload:
    """BASELINE config 5: all-pairs QuotientLoss of an SPD(4) embedding through the fused loss+gradient
    kernel (mm_spd_pdist_loss: no pair vector of distances), pair rows sharded, ONE all-reduce of
    {grad_x, loss, grad_scale}."""

    def __init__(self, d, n, dtype, world, rank, dev, seed=7, comm=None):
        import torch
        from graphembed import _backend as B
        from graphembed.manifolds import SymmetricPositiveDefinite
        from graphembed.objectives import QuotientLoss
        self.d, self.n, self.world, self.rank, self.comm = d, n, world, rank, comm
        self.x, _ = synthetic_spd(n, d, seed, dev, ir=0.1, dtype=dtype, with_g=False)
        self.x.requires_grad_()
        self.scale = torch.tensor(0.5, dtype=dtype, device=dev, requires_grad=True)
        self.rows = B.shard_rows(n, world, rank)
        self.lo, self.hi = B.pair_offset(n, self.rows[0]), B.pair_offset(n, self.rows[1])
        gen = torch.Generator(device=dev).manual_seed(seed + 1000 + rank)
        # stands in for normalised squared graph distances: U(0.01, 1), this rank's slice only
        self.target = torch.rand(self.hi - self.lo, generator=gen, dtype=dtype, device=dev) * 0.99 + 0.01
        self.man = SymmetricPositiveDefinite(d)
        self.spec = QuotientLoss().fused_spec(epoch=3, alpha=1.0)
        self.P = n * (n - 1) // 2
        self.flat = torch.empty(self.x.numel() + 2, dtype=dtype, device=dev)

    def kernels(self):
        import torch
        loss = self.man.pdist_loss(self.x, self.scale, self.target, self.spec, rows=self.rows)
        gx, gs = torch.autograd.grad(loss, (self.x, self.scale))
        # one flat message: the point gradient, the scale gradient and the local loss
        torch.cat([gx.reshape(-1), gs.reshape(1), loss.detach().reshape(1)], out=self.flat)
        return self.flat

    reduce = PdistWorkload.reduce

    def eager_step(self):
        flat = self.kernels()
        if self.world > 1:
            self.reduce(flat)
        return flat


class VecPdistWorkload:
    """BASELINE configs 1 and 2: d2 = man.pdist(x, squared=True); d2.backward(g) on a vector manifold (tree40 ->
    Euclidean R^10; facebook, n = 4039 -> Lorentz H^10: Gram forward / backward on the matrix cores)."""

    def __init__(self, kind, m, n, dtype, dev, seed=0):
        import torch
        from graphembed import manifolds as M
        self.man = {'lorentz': M.Lorentz, 'euclidean': M.Euclidean, 'sphere': M.Sphere}[kind](m)
        torch.manual_seed(seed)
        self.x = self.man.rand(n, out=torch.empty(0, device=dev, dtype=dtype)).requires_grad_()
        self.P = n * (n - 1) // 2
        self.g = torch.randn(self.P, device=dev, dtype=dtype)
        self.n, self.m, self.world, self.rank, self.comm = n, m, 1, 0, None
        self.esz = self.x.element_size()

    def kernels(self):
        import torch
        d2 = self.man.pdist(self.x, squared=True)
        grad, = torch.autograd.grad(d2, self.x, self.g)
        return grad

    eager_step = kernels

    def reduce(self, t):
        return t


class TrainStepWorkload:
    """A full training step of train.py:198-222 — objective (fused loss + gradients), optimizer update of points and
    scales — issued by ONE C-ABI call (mm_train_step_run, graphembed.native_step.NativeTrainStep) and replayed as a graph.
    With a communicator: this rank's pair rows, one all-reduce of {gradients, loss, scale gradients} INSIDE the call."""

    def __init__(self, mans, n, dtype, dev, loss='stress', world=1, rank=0, comm=None, seed=0):
        import torch
        from graphembed.modules import ManifoldEmbedding
        from graphembed.native_step import NativeTrainStep
        from graphembed.objectives import QuotientLoss, StressLoss
        from graphembed.optim import RiemannianSGD
        from graphembed.parallel import PairShard
        torch.manual_seed(seed)
        torch.set_default_dtype(dtype)
        try:
            with torch.device(dev):
                self.emb = ManifoldEmbedding(n, mans)
        finally:
            torch.set_default_dtype(torch.float32)
        self.n, self.world, self.rank, self.comm = n, world, rank, comm
        self.P = n * (n - 1) // 2
        shard = PairShard(n, world=world, rank=rank) if world > 1 else None
        lo, hi = (shard.lo, shard.hi) if shard else (0, self.P)
        gen = torch.Generator(device=dev).manual_seed(seed + 1000 + rank)
        target = torch.rand(hi - lo, generator=gen, dtype=dtype, device=dev) * 0.99 + 0.01
        self.fn = QuotientLoss() if loss == 'quotient' else StressLoss()
        if loss == 'quotient':
            self.fn.on_device(dev)
            self.fn.set_epoch(3, 1.0)
        # (learning rates of 1e-6 / 1e-7: the arithmetic of a step does not depend on them, and the synthetic embedding —
        # random targets — stays in the regime the workload names, the reference's initialisation, for the whole run
        # instead of drifting to wherever a few thousand steps towards random targets take it)
        opts = [RiemannianSGD(list(self.emb.xs), lr=1e-6, exact=True, max_grad_norm=20),
                RiemannianSGD(list(self.emb.scales), lr=1e-7, max_grad_norm=500)]
        self.step = NativeTrainStep(self.emb, self.fn, target, opts, shard=shard, comm=comm if world > 1 else None)
        self.x = self.emb.xs[0]
        self.rows = shard.rows if shard else (0, n)
        self.lo, self.hi = lo, hi
        self.in_call_collective = comm is not None and world > 1

    def kernels(self):
        return self.step(epoch=3, alpha=1.0)

    eager_step = kernels

    def reduce(self, t):
        return t


def time_workload(wl, steps, warmup, fence, use_graph, graph_collective=False, rank=0, tag='', warm_seconds=0.0,
                  min_timed_seconds=0.0):
    """Times `steps` steps of the workload between fences.  Returns (elapsed_s, launch_mode, phases) where
    phases = per-step means, in us, of this rank's device time in the kernels and in the all-reduce and of
    the rest of the step's wall time (host gap), from an instrumented pass after the timed region.
    min_timed_seconds > 0 (secondary workloads only — the headline times EXACTLY the steps it is asked for): the step
    count is raised until the timed region lasts that long (20 steps of a 50-us workload are one millisecond: a single
    scheduling hiccup doubles the figure); phases['steps'] is the count used."""
    import torch
    world = wl.world
    if world > 1:
        import torch.distributed as dist
    graph, static = None, None
    mode = 'eager'
    in_graph_collective = False
    dev = wl.x.device
    # (a workload whose step already contains its collective — the one-call training step with a communicator — has
    # nothing to reduce afterwards: it is captured whole, with the thread-local capture mode RCCL needs)
    in_call = getattr(wl, 'in_call_collective', False)
    if use_graph and in_call:
        try:
            graph, static = graph_of(wl.kernels, fence, with_collective=True)
            ok = True
        except Exception as exc:  # noqa: BLE001
            ok, graph = False, None
            print(f'[bench] rank {rank}: {tag}capture of the one-call step failed ({type(exc).__name__}: {exc}); eager calls',
                  file=sys.stderr)
        if all_agree(ok, world, dev):
            mode, in_graph_collective = 'hipGraph replay (one C-ABI call: kernels + all-reduce + optimizer)', True
        else:
            graph, mode = None, 'eager (one C-ABI call per step)'
            torch.cuda.synchronize()
    elif use_graph:
        # level 1: kernels AND the all-reduce in one captured graph (the host is out of the step entirely);
        # level 2: the kernels captured, the all-reduce issued eagerly behind each replay;
        # level 3 (only if the kernels themselves cannot be captured): eager launches.
        # A level that fails on ANY rank is abandoned by all of them.
        if world > 1 and graph_collective:
            try:
                graph, static = graph_of(wl.eager_step, fence, with_collective=True)
                ok = True
            except Exception as exc:  # noqa: BLE001
                ok, graph = False, None
                print(f'[bench] rank {rank}: {tag}capture with the all-reduce failed ({type(exc).__name__}: {exc})',
                      file=sys.stderr)
            if all_agree(ok, world, dev):
                mode, in_graph_collective = 'hipGraph replay (kernels + all-reduce)', True
            else:
                graph = None
                torch.cuda.synchronize()
        if graph is None:
            try:
                graph, static = graph_of(wl.kernels, fence, with_collective=world > 1)
                ok = True
            except Exception as exc:  # noqa: BLE001 — report and measure eagerly instead
                ok, graph = False, None
                print(f'[bench] rank {rank}: {tag}hipGraph capture unavailable ({type(exc).__name__}: {exc}); eager launches',
                      file=sys.stderr)
            if all_agree(ok, world, dev):
                mode = 'hipGraph replay (kernels)' + (' + eager all-reduce' if world > 1 else '')
            else:
                graph, mode = None, 'eager'
                fence()

    if in_call:
        in_graph_collective = True     # (nothing to issue after the kernels, graphed or not)

    def run():
        if graph is None:
            wl.eager_step()
            return
        graph.replay()
        if world > 1 and not in_graph_collective:
            wl.reduce(static)

    for _ in range(warmup):
        run()
    if warm_seconds > 0:
        # every workload follows seconds of host-side input generation with an idle GPU: a handful of warm-up steps
        # ends before the clocks are back up (a 1.4 ms step measured 2.2 ms), so they also warm up for a minimum TIME
        # (every rank must run the SAME number of steps — each holds a collective: the decision to go on is itself reduced)
        torch.cuda.synchronize()
        tw0 = time.perf_counter()
        while True:
            for _ in range(4):
                run()
            torch.cuda.synchronize()
            more = time.perf_counter() - tw0 < warm_seconds
            if world > 1:
                flag = torch.tensor([1.0 if more else 0.0], device=wl.x.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                more = flag.item() > 0
            if not more:
                break
    if min_timed_seconds > 0:
        fence()
        tp = time.perf_counter()
        for _ in range(steps):
            run()
        fence()
        per = max((time.perf_counter() - tp) / steps, 1e-6)
        want = float(min(5000, max(steps, int(min_timed_seconds / per) + 1)))
        if world > 1:   # every rank holds a collective per step: all of them use the same count
            t = torch.tensor([want], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            want = t.item()
        steps = int(want)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    fence()
    elapsed = time.perf_counter() - t0

    # phase pass (not timed): events on the launch stream around the kernels and around the collective
    k2 = max(3, min(steps, 20))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(k2)]
    fence()
    tw = time.perf_counter()
    for e0, e1, e2 in ev:
        e0.record()
        if graph is None:
            out = wl.kernels()
        else:
            graph.replay()
            out = static
        e1.record()
        if world > 1 and not in_graph_collective:
            wl.reduce(out)
        e2.record()
    fence()
    wall_us = (time.perf_counter() - tw) / k2 * 1e6
    kern_us = median([e0.elapsed_time(e1) * 1e3 for e0, e1, _ in ev])
    coll_us = median([e1.elapsed_time(e2) * 1e3 for _, e1, e2 in ev])
    # SURVEY.md §8d's form of the figure: the MEDIAN of single synchronised steps (each one issued into an idle device and
    # waited for: launch latency and the synchronisation are inside it), next to the pipelined mean the headline reports
    sync_us = []
    for _ in range(max(20, min(steps, 50))):
        fence()
        ts = time.perf_counter()
        run()
        fence()
        sync_us.append((time.perf_counter() - ts) * 1e6)
    phases = {'steps': steps, 'kernels_us': kern_us, 'allreduce_us': coll_us if world > 1 else 0.0,
              'host_gap_us': max(0.0, wall_us - kern_us - (coll_us if world > 1 else 0.0)),
              'step_wall_us': wall_us, 'synchronised_step_us_median': median(sync_us), 'synchronised_steps': len(sync_us)}
    if in_graph_collective:
        phases['note'] = 'all-reduce inside the graph: kernels_us includes it'
    phases['_run'] = run          # (for the caller's kernel-timing pass; removed before the phases are reported)
    return elapsed, mode, phases


def reduce_max(value, dev, world):
    import torch
    t = torch.tensor([value], device=dev, dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def gather_objects(obj, world):
    if world == 1:
        return [obj]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def worker(args):
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    # a hung rank (lost peer, wedged collective) must not hang the job: hard exit, non-zero
    def expire():
        print(f'[bench] rank {rank}: still running after --rank-timeout {args.rank_timeout:.0f} s; exiting 3',
              file=sys.stderr, flush=True)
        os._exit(3)
    watchdog = threading.Timer(args.rank_timeout, expire)
    watchdog.daemon = True
    watchdog.start()

    import torch
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit('bench.py needs an MI355X (no GPU visible); there is no CPU path in the product')
    backend = os.environ.get('MM_BENCH_BACKEND', 'nccl')
    if world > 1 and backend == 'nccl' and ndev < world:
        raise SystemExit(f'--gpus {world} but {ndev} GPU(s) visible: one rank per GPU is required with RCCL '
                         '(MM_BENCH_BACKEND=gloo shares a GPU between ranks — a dry run of the N > 1 code path)')
    dev = torch.device('cuda', local_rank % ndev)
    torch.cuda.set_device(dev)
    if world > 1:
        import datetime
        import torch.distributed as dist
        # "nccl" IS RCCL on ROCm.  MM_BENCH_BACKEND=gloo exists only to exercise the N > 1 code
        # path on a single-GPU development box.
        tmo = datetime.timedelta(seconds=min(args.rank_timeout, 300.0))
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)

    from graphembed import _backend as B
    lib = B.lib()
    fence = Fence(world)
    use_graph = args.graph in ('on', 'auto')
    # N > 1 over RCCL: the collective goes through the library's own communicator (mm_comm_init / mm_allreduce_sum,
    # one per process, created once) and is captured in the step's graph by default; MM_BENCH_GRAPH_COLLECTIVE=0 keeps
    # it outside (graph-replayed kernels + eager all-reduce).  Ranks that share a GPU (gloo dry run) cannot build an
    # RCCL communicator: they reduce through torch.distributed, outside the graph.
    comm = None
    if world > 1 and backend == 'nccl':
        from graphembed.comm import Communicator
        comm = Communicator.from_torch_distributed(dev)
    graph_collective = comm is not None and os.environ.get('MM_BENCH_GRAPH_COLLECTIVE', '1') != '0'
    n = args.n

    # ---- headline ---------------------------------------------------------------------------
    wl = PdistWorkload(DIM, n, torch.float32, 0.1, world, rank, dev, comm=comm)
    # (W warm-up steps as asked, then 50 ms more of untimed steps: W = 10 steps are under a millisecond of GPU time, not
    # enough for the clocks to settle after the seconds of host-side input generation)
    elapsed, mode, phases = time_workload(wl, args.steps, args.warmup, fence, use_graph, graph_collective, rank,
                                          warm_seconds=0.05)
    elapsed = reduce_max(elapsed, dev, world)

    # per-kernel durations: HIP events attached to each pair kernel's own dispatch on the launch stream
    # (hipExtLaunchKernelGGL start / stop events, csrc/prof.hpp: the kernel's execution time as a profiler sees it).
    # They cannot live inside a captured graph, so they time the kernels of an eager pass of the same step issued
    # right after the timed region; one discarded pass first (event creation, allocator warm-up).
    kern, eager_ms, eager_st_ms = {'fwd': None, 'bwd': None}, None, None
    run_step = phases.pop('_run')
    if not args.no_prof:
        # (a) the same step through Python autograd, one synchronised step at a time: its host cost
        wl.eager_step()
        fence()
        times = []
        for _ in range(max(5, min(args.steps, 20))):
            te = time.perf_counter()
            wl.eager_step()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - te)
        fence()
        eager_ms = median(times) * 1e3
        # ... and with the autograd engine's worker threads off (torch.autograd.set_multithreading_enabled(False): the
        # backward node runs on the calling thread, no hand-off) — what a caller of the eager plugin API can switch on itself
        eager_st_ms = None
        if hasattr(torch.autograd, 'set_multithreading_enabled'):
            with torch.autograd.set_multithreading_enabled(False):
                wl.eager_step()
                torch.cuda.synchronize()
                times = []
                for _ in range(max(5, min(args.steps, 20))):
                    te = time.perf_counter()
                    wl.eager_step()
                    torch.cuda.synchronize()
                    times.append(time.perf_counter() - te)
            fence()
            eager_st_ms = median(times) * 1e3
        # (b) kernel durations in the clock regime of the timed region: each profiled eager step is queued behind a
        # burst of replays of the timed step, with no synchronisation in between — the device never idles (issued one
        # step at a time from Python it idles 45 % of the time, the clocks drop and the kernels take ~10 % longer)
        lib.call('mm_prof_enable', 1)
        wl.eager_step()
        fence()
        collect_kernel_us(lib)                 # discard
        for _ in range(max(5, min(args.steps, 20))):
            for _ in range(8):
                run_step()
            wl.eager_step()
        fence()
        lib.call('mm_prof_enable', 0)
        kern = collect_kernel_us(lib)
    clock_mhz = None
    if not args.no_prof:
        try:
            clock_mhz = shader_clock_mhz(lib, run_step, dev)
        except Exception as exc:  # noqa: BLE001 — a diagnostic: never fails the run
            print(f'[bench] shader clock probe failed ({type(exc).__name__}: {exc})', file=sys.stderr)
    per_rank = gather_objects({'rank': rank, 'rows': list(wl.rows), 'pairs': wl.hi - wl.lo, **phases,
                               'fwd_kernel_us': kern['fwd'], 'bwd_kernel_us': kern['bwd']}, world)

    out = None
    if rank == 0:
        P = wl.P
        pairs_local = wl.hi - wl.lo
        esz = 4
        out = {
            'metric': 'pairwise manifold-dist/sec (fwd+bwd), SPD(3) 5k-node',
            'value': P * args.steps / elapsed, 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'grqc-class graph, n={n} nodes -> SPD(3) affine-invariant, all '
                                   f'{P} pairs, squared distance + backward, reference init (||log X||=0.1)',
                       'pairs_per_step': P, 'parallelism': f'pair-rows sharded x{world}, 1 all-reduce',
                       'launch': mode, 'backend': backend if world > 1 else None,
                       'collective': (None if world == 1 else
                                      'mm_allreduce_sum (C ABI, process-lifetime RCCL communicator)' if comm is not None
                                      else f'torch.distributed {backend}'),
                       'untimed_warm_seconds': 0.05},
            'per_rank': per_rank,
            'kernel_source_hash': kernel_source_hash(),
            # the same step measured SURVEY.md §8d's way: median of single synchronised steps (launch latency + sync inside)
            'ms_per_step_median_synchronised': (phases.get('synchronised_step_us_median') or 0.0) / 1e3 or None,
        }
        if eager_ms is not None:
            # same step issued through Python autograd (host-bound); median of single synchronised steps
            out['eager_ms_per_step'] = eager_ms
            from graphembed import _backend as _B
            out['eager_host'] = ('C++ autograd nodes (lib/_mm_autograd.so)' if _B.autograd_ext() is not None
                                 else 'torch.autograd.Function classes')
            if eager_st_ms is not None:
                out['eager_ms_per_step_autograd_single_thread'] = eager_st_ms
        pmc = stamped_pmc() if (n == N_NODES and world == 1) else None
        if kern['bwd']:
            # dominant kernel: spd_pdist_bwd.  Algorithmic HBM bytes per launch: read g (4 B per pair) + node
            # factors in (2*6 floats) + accumulators out (2*6 floats) per node.
            by = pairs_local * esz + n * 24 * esz
            t = kern['bwd'] * 1e-6
            rk = (pmc or {}).get('kernels', {}).get('spd_pdist_bwd', {})
            out['roofline'] = {'bound': 'hbm', 'kernel': 'spd_pdist_bwd_kernel<float,3,...>',
                               'achieved': by / t / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': by / t / 1e9 / HBM_PEAK_GBS,
                               'traffic': rk.get('traffic_bytes'),
                               'traffic_source': (pmc or {}).get('source') if rk else None,
                               'algorithmic_bytes': by, 'avg_launch_us': kern['bwd'],
                               'measured_in': 'eager steps queued behind replays of the timed step right after the timed region (device never idle); HIP events attached to the kernel dispatch on the launch stream (hipExtLaunchKernelGGL)',
                               'valu_insts_per_64_pairs': rk.get('valu_insts_per_64_pairs')}
            # the honest ceiling (SURVEY.md §8d "vector-ALU"): these kernels issue one instruction of a wavefront
            # per ~2 cycles per SIMD whatever its kind, so issue time = instructions per wavefront-row x rows / SIMDs x 2 cycles
            # at the shader clock of THIS run; the fraction of the kernel's duration it fills is how close the kernel is to
            # that machine limit.  Instruction count: the stamped counters of this very kernel build (null when stale).
            insts = rk.get('insts_any_per_64_pairs')
            if insts and clock_mhz:
                issue_us = insts * (pairs_local / 64.0) / N_SIMD * ISSUE_CYCLES / clock_mhz
                out['roofline'].update({'valu_issue_frac': issue_us / kern['bwd'], 'issue_time_us': issue_us,
                                        'insts_per_64_pairs': insts, 'shader_clock_mhz': clock_mhz,
                                        'issue_model': f'{insts:.0f} instructions per 64 pairs x {pairs_local / 64.0:.0f} wavefront-rows / {N_SIMD} SIMDs x {ISSUE_CYCLES:.0f} cycles / clock'})
            else:
                out['roofline'].update({'valu_issue_frac': None, 'shader_clock_mhz': clock_mhz})
            if kern['fwd']:
                byf = pairs_local * esz + n * 12 * esz
                tf = kern['fwd'] * 1e-6
                rf = (pmc or {}).get('kernels', {}).get('spd_pdist_fwd', {})
                out['roofline_fwd'] = {'bound': 'hbm', 'kernel': 'spd_pdist_fwd_kernel<float,3,...>',
                                       'achieved': byf / tf / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                       'frac': byf / tf / 1e9 / HBM_PEAK_GBS, 'traffic': rf.get('traffic_bytes'),
                                       'algorithmic_bytes': byf, 'avg_launch_us': kern['fwd'],
                                       'valu_insts_per_64_pairs': rf.get('valu_insts_per_64_pairs')}
    del wl
    torch.cuda.empty_cache()

    # ---- secondary workloads, same JSON line ----------------------------------------------------
    extra = []
    if not args.no_extra:
        kbase, w2 = max(5, min(args.steps, 20)), max(2, min(args.warmup, 5))
        k2 = kbase   # (per workload: the count its timed region used, >= kbase — time_workload, min_timed_seconds)
        cases = []
        if world == 1 and n == N_NODES:
            cases += [('SPD(3) n=5000 f32, mid-training spread (||log X||=0.35)', DIM, N_NODES, torch.float32, 0.35),
                      ('SPD(3) n=5000 f64 (the dtype run.py:32-35 sets), reference init', DIM, N_NODES, torch.float64, 0.1),
                      ('SPD(3) n=5000 f64, mid-training spread (||log X||=0.35)', DIM, N_NODES, torch.float64, 0.35),
                      ('SPD(3) n=4158 f32 (grqc, BASELINE config 3), reference init', DIM, 4158, torch.float32, 0.1),
                      ('SPD(4) n=2274 f32 (BASELINE config 5, small graph), reference init', 4, 2274, torch.float32, 0.1)]
        for name, d, nn, dt, ir in cases:
            w = PdistWorkload(d, nn, dt, ir, world, rank, dev, comm=comm)
            el, md, ph = time_workload(w, kbase, w2, fence, use_graph, graph_collective, rank, tag=name + ': ', warm_seconds=0.05, min_timed_seconds=0.02)
            k2 = ph['steps']
            el = reduce_max(el, dev, world)
            run_w = ph.pop('_run')
            kk = {'fwd': None, 'bwd': None}
            if not args.no_prof:
                lib.call('mm_prof_enable', 1)
                w.eager_step()
                fence()
                collect_kernel_us(lib)
                for _ in range(5):
                    for _ in range(8):
                        run_w()
                    w.eager_step()
                fence()
                lib.call('mm_prof_enable', 0)
                kk = collect_kernel_us(lib)
            rec = {'workload': name + ', pdist fwd+bwd', 'pairs_per_step': w.P, 'ms_per_step': el / k2 * 1e3,
                   'value': w.P * k2 / el, 'unit': 'pairs/s', 'steps': k2, 'launch': md,
                   'fwd_kernel_us': kk['fwd'], 'bwd_kernel_us': kk['bwd']}
            if kk['bwd']:
                np_ = d * (d + 1) // 2
                byb = (w.hi - w.lo) * w.esz + nn * 4 * np_ * w.esz
                rec['bwd_hbm_frac'] = byb / (kk['bwd'] * 1e-6) / 1e9 / HBM_PEAK_GBS
                tname = 'double' if dt == torch.float64 else 'float'
                rec['roofline'] = {'bound': 'hbm', 'kernel': f'spd_pdist_bwd_kernel<{tname},{d},...>', 'achieved': byb / (kk['bwd'] * 1e-6) / 1e9,
                                   'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': rec['bwd_hbm_frac'], 'traffic': None,
                                   'algorithmic_bytes': byb, 'avg_launch_us': kk['bwd']}
                if kk['fwd']:
                    byf = (w.hi - w.lo) * w.esz + nn * 2 * np_ * w.esz
                    rec['roofline_fwd'] = {'bound': 'hbm', 'kernel': f'spd_pdist_fwd_kernel<{tname},{d},...>', 'achieved': byf / (kk['fwd'] * 1e-6) / 1e9,
                                           'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': byf / (kk['fwd'] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                           'traffic': None, 'algorithmic_bytes': byf, 'avg_launch_us': kk['fwd']}
            if rank == 0 and out is not None and dt == torch.float64 and nn == N_NODES and 'roofline' in rec:
                # the headline workload in the dtype run.py:32-35 sets: its own roofline blocks on the JSON line
                key = 'roofline_f64' if ir == 0.1 else 'roofline_f64_mid_training'
                out[key] = dict(rec['roofline'], workload=name, ms_per_step=rec['ms_per_step'],
                                fwd=rec.get('roofline_fwd'))
            extra.append(rec)
            del w
            torch.cuda.empty_cache()
        if world == 1 and n == N_NODES:
            from graphembed import manifolds as M
            # BASELINE configs 2 and 1 (pdist fwd + bwd on vector manifolds; Lorentz: Gram forward on the matrix cores, symmetric VALU backward)
            for name, kind, m, nn, dt in (('BASELINE config 2: facebook-class graph n=4039 -> Lorentz H^10 (11 coords) f32', 'lorentz', 11, 4039, torch.float32),
                                          ('BASELINE config 2 in f64 (the dtype run.py:32-35 sets)', 'lorentz', 11, 4039, torch.float64),
                                          ('BASELINE config 1: tree40 n=40 -> Euclidean R^10 f64 (plumbing: launch-bound)', 'euclidean', 10, 40, torch.float64)):
                w = VecPdistWorkload(kind, m, nn, dt, dev)
                el, md, ph = time_workload(w, kbase, w2, fence, use_graph, False, rank, tag=name + ': ', warm_seconds=0.05, min_timed_seconds=0.02)
                k2 = ph['steps']
                run_w = ph.pop('_run')
                kk = {'fwd': None, 'bwd': None}
                if not args.no_prof:
                    lib.call('mm_prof_enable', 1)
                    w.eager_step()
                    fence()
                    collect_kernel_us(lib, (('fwd', 2), ('bwd', 3)))
                    for _ in range(5):
                        for _ in range(8):
                            run_w()
                        w.eager_step()
                    fence()
                    lib.call('mm_prof_enable', 0)
                    kk = collect_kernel_us(lib, (('fwd', 2), ('bwd', 3)))
                rec = {'workload': name + ', pdist fwd+bwd', 'pairs_per_step': w.P, 'ms_per_step': el / k2 * 1e3,
                       'value': w.P * k2 / el, 'unit': 'pairs/s', 'steps': k2, 'launch': md,
                       'fwd_kernel_us': kk['fwd'], 'bwd_kernel_us': kk['bwd']}
                if kk['bwd']:   # algorithmic bytes: read g (one element per pair) + points in + gradient out
                    rec['bwd_hbm_frac'] = (w.P * w.esz + 2 * nn * m * w.esz) / (kk['bwd'] * 1e-6) / 1e9 / HBM_PEAK_GBS
                if kk['fwd']:
                    rec['fwd_hbm_frac'] = (w.P * w.esz + nn * m * w.esz) / (kk['fwd'] * 1e-6) / 1e9 / HBM_PEAK_GBS
                extra.append(rec)
                del w
                torch.cuda.empty_cache()
            # full training steps (objective + optimizer; ONE C-ABI call, replayed as a graph): configs 3, 4 and 2
            for name, mans, nn, dt, loss in (
                    ('BASELINE config 3 training step: SPD(3) n=5000 f32, StressLoss + RSGD (2 launches: pair kernel, fused finalize+update+tables)',
                     lambda: [M.SymmetricPositiveDefinite(3)], N_NODES, torch.float32, 'stress'),
                    ('BASELINE config 4 training step: csphd n=1025 -> H^5 x S^5 x SPD(2) f32, StressLoss + RSGD (2 launches: mixed-manifold pair kernel, fused gradients+updates)',
                     lambda: [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float32, 'stress'),
                    ('BASELINE config 4 training step in f64', lambda: [M.Lorentz(6), M.Sphere(6), M.SymmetricPositiveDefinite(2)], 1025, torch.float64, 'stress'),
                    ('BASELINE config 2 training step: Lorentz(11) n=4039 f32, StressLoss + RSGD (2 launches: symmetric pair kernel, fused gradient+update)', lambda: [M.Lorentz(11)], 4039, torch.float32, 'stress'),
                    ('BASELINE config 5 training step: SPD(4) n=16384 f32, QuotientLoss + RSGD', lambda: [M.SymmetricPositiveDefinite(4)], 16384, torch.float32, 'quotient')):
                w = TrainStepWorkload(mans(), nn, dt, dev, loss=loss)
                el, md, ph = time_workload(w, kbase, w2, fence, use_graph, False, rank, tag=name + ': ', warm_seconds=0.05, min_timed_seconds=0.02)
                k2 = ph['steps']
                ph.pop('_run')
                extra.append({'workload': name, 'pairs_per_step': w.P, 'ms_per_step': el / k2 * 1e3, 'value': w.P * k2 / el,
                              'unit': 'pairs/s', 'steps': k2, 'launch': md + ' of mm_train_step_run'})
                del w
                torch.cuda.empty_cache()
        if world > 1 and n == N_NODES:
            # WEAK scaling of the same path beside the strong-scaling headline: per-GPU work fixed at the headline's 12.5 M
            # pairs — N ranks embed a graph of 5000 sqrt(N) nodes (N x 12.5 M pairs), pair rows sharded, one all-reduce of the
            # (sqrt(N) times larger) gradient.  The 5000-node problem itself is 80 us of work: its strong scaling is bounded
            # by the latency of one collective, this block shows what the design does when a rank has a full launch to chew on.
            nw = int(round(N_NODES * world ** 0.5))
            w = PdistWorkload(DIM, nw, torch.float32, 0.1, world, rank, dev, local_g=True, comm=comm)
            el, md, ph = time_workload(w, kbase, w2, fence, use_graph, graph_collective, rank, tag='weak scaling: ', warm_seconds=0.05, min_timed_seconds=0.02)
            k2 = ph['steps']
            ph.pop('_run')
            el = reduce_max(el, dev, world)
            ranksw = gather_objects({'rank': rank, 'rows': list(w.rows), 'pairs': w.hi - w.lo, **ph}, world)
            extra.append({'workload': f'weak scaling of the headline path: SPD(3) f32, n = 5000 sqrt(N) = {nw} nodes, {w.P} pairs '
                                      f'({w.P // world} per GPU), pdist fwd+bwd, pair rows sharded, 1 all-reduce',
                          'pairs_per_step': w.P, 'ms_per_step': el / k2 * 1e3, 'value': w.P * k2 / el, 'unit': 'pairs/s',
                          'steps': k2, 'n_gpus': world, 'scaling': 'weak', 'launch': md, 'per_rank': ranksw})
            del w
            torch.cuda.empty_cache()
        # the size where sharding pays: BASELINE config 5 (bio-wormnet-class, ~16k nodes, SPD(4), distortion loss)
        w = FusedLossWorkload(4, 16384, torch.float32, world, rank, dev, comm=comm)
        el, md, ph = time_workload(w, kbase, w2, fence, use_graph, graph_collective, rank, tag='config 5: ', warm_seconds=0.1, min_timed_seconds=0.02)
        k2 = ph['steps']
        ph.pop('_run')
        el = reduce_max(el, dev, world)
        ranks5 = gather_objects({'rank': rank, 'rows': list(w.rows), 'pairs': w.hi - w.lo, **ph}, world)
        extra.append({'workload': 'BASELINE config 5: n=16384 nodes -> SPD(4), all-pairs QuotientLoss, fused '
                                  'loss+gradient kernel, pair rows sharded, 1 all-reduce of {grad, loss, grad_scale}',
                      'pairs_per_step': w.P, 'ms_per_step': el / k2 * 1e3, 'value': w.P * k2 / el, 'unit': 'pairs/s',
                      'steps': k2, 'n_gpus': world, 'scaling': 'strong', 'launch': md, 'per_rank': ranks5})
        del w
        torch.cuda.empty_cache()
        if world > 1 and comm is not None:
            # the same configuration as a full training step through ONE C-ABI call per step: this rank's pair rows ->
            # all-reduce (mm_allreduce_sum inside mm_train_step_run) -> identical RSGD update on every rank
            from graphembed import manifolds as M
            w = TrainStepWorkload([M.SymmetricPositiveDefinite(4)], 16384, torch.float32, dev, loss='quotient', world=world,
                                  rank=rank, comm=comm)
            el, md, ph = time_workload(w, kbase, w2, fence, use_graph, graph_collective, rank, tag='config 5 step: ', warm_seconds=0.1, min_timed_seconds=0.02)
            k2 = ph['steps']
            ph.pop('_run')
            el = reduce_max(el, dev, world)
            ranks5 = gather_objects({'rank': rank, 'rows': list(w.rows), 'pairs': w.hi - w.lo, **ph}, world)
            extra.append({'workload': 'BASELINE config 5 training step: n=16384 -> SPD(4), QuotientLoss + RSGD, pair rows sharded, '
                                      'objective -> all-reduce -> update in ONE mm_train_step_run per rank',
                          'pairs_per_step': w.P, 'ms_per_step': el / k2 * 1e3, 'value': w.P * k2 / el, 'unit': 'pairs/s',
                          'steps': k2, 'n_gpus': world, 'scaling': 'strong', 'launch': md, 'per_rank': ranks5})
            del w
            torch.cuda.empty_cache()

    if rank == 0:
        if extra:
            out['extra'] = extra
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(42, n)
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.destroy()
    if world > 1:
        dist.destroy_process_group()
    watchdog.cancel()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and 'RANK' not in os.environ:
        launch(args, argv)       # does not return
    worker(args)


if __name__ == '__main__':
    main()
