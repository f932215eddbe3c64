"""Riemannian SGD over the Manifold API — counterpart of
graphembed/graphembed/optim/rsgd.py:10-82 (same param-group keys and update order:
egrad2rgrad -> per-point norm clip -> momentum/transport or plain exp|retr step)."""
import torch
from torch.optim.optimizer import required

from graphembed.modules import ManifoldParameter


class _Flat:
    """Flat parameters (scales, curvatures): the reference falls back to Euclidean(1)
    (rsgd.py:7,56-59); the arithmetic is trivial and stays in torch."""

    def egrad2rgrad(self, x, u):
        return u

    def norm(self, x, u, keepdim=False):
        return (u * u).sum(-1, keepdim=keepdim).clamp(min=1e-8).sqrt()

    def exp(self, x, u):
        return x + u

    retr = exp

    def transp(self, x, y, u):
        return u

    def rsgd_step(self, x, egrad, *, lr, max_grad_norm=None, exact=False):
        """One fused kernel (the Euclidean RSGD step over the last dimension) instead of ~12 scalar
        framework kernels per parameter; None on CPU tensors (the torch path above is used)."""
        if not x.is_cuda:
            return None
        from graphembed import _backend as B
        m = x.shape[-1] if x.ndim else 1
        if m > 32 or x.dtype not in (torch.float32, torch.float64):
            return None
        xc = x.detach().reshape(-1, m).contiguous()
        gc = egrad.detach().reshape(-1, m).to(xc.dtype).contiguous()
        with B.on_device(xc.device):
            out = torch.empty_like(xc)
            B.lib().call('mm_vec_rsgd_step', B.dtype_code(xc), B.EUCLIDEAN, B.ptr(xc), B.ptr(gc),
                         xc.shape[0], m, float(lr),
                         -1.0 if max_grad_norm is None else float(max_grad_norm), int(bool(exact)),
                         B.ptr(out), B.stream_of(xc))
        return out.reshape(x.shape)


_default_manifold = _Flat()


def _assign(x, new):
    """`x.set_(new)` as in the reference (rsgd.py:80-82); while a HIP graph is being captured the
    value is copied into x's own storage instead, so a replayed step keeps advancing the same
    parameter memory."""
    if x.is_cuda and torch.cuda.is_current_stream_capturing():
        x.copy_(new)
    else:
        x.set_(new)


class RiemannianSGD(torch.optim.Optimizer):
    graph_safe = True  # no host-side per-step state: a captured step can be replayed (graphembed.graphed)

    def __init__(self, params, lr=required, momentum=0, dampening=0, max_grad_norm=None,
                 exact=False):
        if momentum < 0.0:
            raise ValueError('Invalid momentum value: {}'.format(momentum))
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening,
                        max_grad_norm=max_grad_norm, exact=exact)
        super().__init__(params, defaults)

    def step(self, closure=None):
        loss = None
        if closure is not None:
            loss = closure()
        with torch.no_grad():
            for group in self.param_groups:
                self._step(group)
        return loss

    # torch.optim.Optimizer wraps `step` of every subclass in a profiler range plus pre/post hook
    # dispatch (~25 us of host time per call) unless it is marked as hooked already: the update is
    # one or two kernels of a few microseconds, so the wrapper alone would dominate an eager step.
    # (Step hooks registered on the optimizer are therefore not run.)
    step.hooked = True

    def zero_grad(self, set_to_none=True):
        """Lean version of Optimizer.zero_grad (same semantics; the foreach/profiler machinery of
        the base class costs ~20 us per call for a handful of parameters)."""
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                if set_to_none:
                    p.grad = None
                else:
                    if p.grad.grad_fn is not None:
                        p.grad.detach_()
                    else:
                        p.grad.requires_grad_(False)
                    p.grad.zero_()

    def _step(self, group):
        lr, momentum, dampening = group['lr'], group['momentum'], group['dampening']
        max_grad_norm = group['max_grad_norm']
        for x in group['params']:
            grad = x.grad
            if grad is None:
                continue
            state = self.state[x]
            if len(state) == 0 and momentum > 0:
                state['momentum_buffer'] = grad.clone()
            if isinstance(x, ManifoldParameter) and x.manifold is not None:
                manifold = x.manifold
            else:
                manifold = _default_manifold

            # one fused kernel per parameter when the manifold offers it
            fused = getattr(manifold, 'rsgd_step', None)
            if momentum == 0 and fused is not None:
                new_x = fused(x, grad, lr=lr, max_grad_norm=max_grad_norm, exact=group['exact'])
                if new_x is not None:
                    _assign(x, new_x)
                    continue

            retr = manifold.exp if group['exact'] else manifold.retr
            grad = manifold.egrad2rgrad(x, grad)
            if max_grad_norm is not None:
                grad_norm = manifold.norm(x, grad, keepdim=True)
                grad = grad * torch.clamp(max_grad_norm / grad_norm, max=1.0)
            if momentum > 0:
                buf = state['momentum_buffer']
                buf.mul_(momentum).add_(grad, alpha=1 - dampening)
                new_x = retr(x, -lr * buf)
                new_buf = manifold.transp(x, new_x, buf)
                _assign(x, new_x)
                _assign(buf, new_buf)
            else:
                _assign(x, retr(x, -lr * grad))
