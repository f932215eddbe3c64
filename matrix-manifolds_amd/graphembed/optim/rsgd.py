"""Riemannian SGD over the Manifold API — counterpart of
graphembed/graphembed/optim/rsgd.py:10-82: same constructor and param-group keys, same update
order (egrad2rgrad -> per-point norm clip -> momentum with transport | plain exp/retr step)."""
from torch.optim.optimizer import required

from graphembed.optim._common import FLAT, ManifoldOptimizer, assign, capturing

_default_manifold = FLAT  # (kept under the reference's name for callers that import it)
_assign = assign


class RiemannianSGD(ManifoldOptimizer):
    graph_safe = True  # no host-side per-step state: a captured step can be replayed (graphembed.graphed)

    def __init__(self, params, lr=required, momentum=0, dampening=0, max_grad_norm=None,
                 exact=False):
        if momentum < 0.0:
            raise ValueError('Invalid momentum value: {}'.format(momentum))
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening,
                                      max_grad_norm=max_grad_norm, exact=exact))

    def _update(self, group, p, state, manifold):
        lr, momentum, clip = group['lr'], group['momentum'], group['max_grad_norm']
        if momentum == 0:
            # one fused kernel per parameter when the manifold offers it (rsgd.py:63-68,82)
            fused = getattr(manifold, 'rsgd_step', None)
            if fused is not None and capturing(p):
                # while a HIP graph is recorded the kernel writes straight over the parameter
                if fused(p, p.grad, lr=lr, max_grad_norm=clip, exact=group['exact'], inplace=True) is p:
                    return
            new_p = None if fused is None else fused(p, p.grad, lr=lr, max_grad_norm=clip,
                                                     exact=group['exact'])
            if new_p is None:
                rgrad, _ = self.riemannian_gradient(manifold, p, clip)
                move = manifold.exp if group['exact'] else manifold.retr
                new_p = move(p, -lr * rgrad)
            assign(p, new_p)
            return
        # heavy-ball momentum, transported to the new point (rsgd.py:70-80)
        if 'momentum_buffer' not in state:
            state['momentum_buffer'] = p.grad.clone()
        rgrad, _ = self.riemannian_gradient(manifold, p, clip)
        buf = state['momentum_buffer']
        buf.mul_(momentum).add_(rgrad, alpha=1 - group['dampening'])
        move = manifold.exp if group['exact'] else manifold.retr
        new_p = move(p, -lr * buf)
        carried = manifold.transp(p, new_p, buf)
        assign(p, new_p)
        assign(buf, carried)
