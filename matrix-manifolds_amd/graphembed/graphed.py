"""A whole training step (loss, backward, optimizer updates) captured once as a HIP graph and
replayed — the launch-bound regime of the reference's small configurations (n ~ 1e3 nodes, several
factor manifolds: ~60 kernel launches of a few microseconds each per step, train.py:198-222).

Everything on the path is capture-safe: the HIP entry points take the stream, allocate nothing and
never synchronise; workspaces and outputs come from torch's caching allocator (graph-private pool
during capture); `RiemannianSGD` / `RiemannianAdam` write parameters in place while capturing.

    step = GraphedTrainStep(lambda: objective(None, epoch=0, alpha=1.0), optimizers)
    for epoch in range(n_epochs):
        loss = step()            # one hipGraphLaunch; `loss` is a device scalar (no sync)

Anything that changes between steps must live in device memory that the closure reads (targets,
learning-rate tensors); Python scalars are frozen at capture time — re-capture (`step.capture()`)
when they change (e.g. the quotient loss's `epoch`)."""
import torch


class GraphedTrainStep:

    def __init__(self, loss_fn, optimizers, warmup=3):
        self.loss_fn = loss_fn
        self.optimizers = list(optimizers)
        for o in self.optimizers:
            if not getattr(o, 'graph_safe', False):
                raise TypeError(f'{type(o).__name__} keeps host-side step state and cannot be replayed '
                                'from a captured graph (RiemannianSGD can)')
        self.warmup = warmup
        self.graph = None
        self.loss = None

    def _params(self):
        return [p for o in self.optimizers for g in o.param_groups for p in g['params']]

    def _eager_step(self):
        for o in self.optimizers:
            o.zero_grad(set_to_none=False)
        loss = self.loss_fn()
        loss.backward()
        for o in self.optimizers:
            o.step()
        return loss.detach()

    def capture(self):
        params = self._params()
        if not params or not all(p.is_cuda for p in params):
            raise RuntimeError('GraphedTrainStep needs parameters in GPU memory')
        for p in params:  # static gradient buffers: zero_grad(set_to_none=False) keeps them
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        # parameters are restored after the warm-up iterations: capture must not advance training
        saved = [p.detach().clone() for p in params]
        states = [o.state_dict() for o in self.optimizers]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for p, s in zip(params, saved):
                p.copy_(s)
        for o, st in zip(self.optimizers, states):
            o.load_state_dict(st)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._eager_step()
        # the capture pass itself does not execute: parameters are still the saved ones
        return self

    def __call__(self):
        if self.graph is None:
            self.capture()
        self.graph.replay()
        return self.loss
