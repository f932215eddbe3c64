"""A whole training step (loss, backward, optimizer updates) captured once as a HIP graph and
replayed — the launch-bound regime of the reference's small configurations (n ~ 1e3 nodes, several
factor manifolds: ~60 kernel launches of a few microseconds each per step, train.py:198-222).

Everything on the path is capture-safe: the HIP entry points take the stream, allocate nothing and
never synchronise; workspaces and outputs come from torch's caching allocator (graph-private pool
during capture); `RiemannianSGD` / `RiemannianAdam` write parameters in place while capturing and keep
all per-step state (momentum, moments, the Adam step counter) in device memory.

    step = GraphedTrainStep(lambda: objective(None, epoch=0, alpha=1.0), optimizers).capture()
    # capture() has run `warmup` (default 3) ordinary steps: step.warmup_losses
    for epoch in range(3, n_epochs):
        loss = step()            # one hipGraphLaunch; `loss` is a device scalar (no sync)

Anything that changes between steps must live in device memory that the closure reads (index buffers,
targets; the quotient loss's {alpha, eps}: `QuotientLoss.on_device`).  Optimizer hyper-parameters are
by-value kernel arguments: when one changes (a learning-rate scheduler) the next call notices, runs that
step eagerly and records the graph again."""
import torch

from graphembed._backend import unit_seed


class GraphedTrainStep:

    def __init__(self, loss_fn, optimizers, warmup=3, unroll=1):
        self.loss_fn = loss_fn
        self.optimizers = list(optimizers)
        for o in self.optimizers:
            if not getattr(o, 'graph_safe', False):
                raise TypeError(f'{type(o).__name__} keeps host-side step state and cannot be replayed '
                                'from a captured graph (RiemannianSGD and RiemannianAdam of this package can)')
        self.warmup = warmup
        # `unroll` consecutive steps per recorded graph: one hipGraphLaunch (and its ~8 us of launch latency)
        # per `unroll` steps — for steps that are a few tens of microseconds long and need nothing from the host
        # in between (full batch, fixed targets); `__call__` then advances `unroll` steps and returns the last loss
        self.unroll = max(1, int(unroll))
        self.losses = []
        self.graph = None
        self.loss = None
        self.warmup_losses = []

    def _params(self):
        return [p for o in self.optimizers for g in o.param_groups for p in g['params']]

    def _eager_step(self):
        # gradients are dropped, not zeroed: backward then ASSIGNS them (no fill and no accumulate
        # kernel per parameter — 2 launches per parameter of a step that has ~20 in total); while
        # capturing they land in the graph's private pool, where every replay rewrites them in place
        for o in self.optimizers:
            o.zero_grad(set_to_none=True)
        loss = self.loss_fn()
        loss.backward(unit_seed(loss))  # no ones_like fill; the fused objectives skip their `* 1`
        for o in self.optimizers:
            o.step()
        return loss.detach()

    def _hyper(self):
        """The by-value hyper-parameters a recorded step has baked in (learning rates, betas, clip norms ...)."""
        return tuple(tuple(sorted((k, v) for k, v in g.items()
                                  if k != 'params' and isinstance(v, (int, float, bool, tuple, type(None)))))
                     for o in self.optimizers for g in o.param_groups)

    def capture(self, warmup=None):
        """Runs `warmup` ordinary (eager) training steps — they COUNT as training steps: optimizer
        state is created and advanced by them, and their losses are kept in `warmup_losses` — then
        records the next step without executing it."""
        params = self._params()
        if not params or not all(p.is_cuda for p in params):
            raise RuntimeError('GraphedTrainStep needs parameters in GPU memory')
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.warmup_losses = [self._eager_step() for _ in range(max(1, self.warmup if warmup is None else warmup))]
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.losses = [self._eager_step() for _ in range(self.unroll)]
            self.loss = self.losses[-1]
        self._recorded = self._hyper()
        return self

    def __call__(self):
        if self.graph is None:
            self.capture()
        elif self._recorded != self._hyper():
            # a scheduler changed a learning rate (ReduceLROnPlateau, train.py:174): this step runs eagerly
            # with the new values and the graph is recorded again for the following ones
            self.capture(warmup=1)
            return self.warmup_losses[-1]
        self.graph.replay()
        return self.loss
