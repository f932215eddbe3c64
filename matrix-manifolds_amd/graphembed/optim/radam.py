"""Riemannian Adam with one second-moment scalar per point — counterpart of
graphembed/graphembed/optim/radam.py:12-98 (same constructor, param groups, state keys and update
order), built on the shared scaffold of `_common.py`.

Unlike the reference, the step counter lives in device memory and the bias corrections are computed
there, so a captured HIP graph of a training step replays correctly (`graph_safe`)."""
import logging
import os

import torch

from graphembed.optim._common import ManifoldOptimizer, assign, capturing
from graphembed.utils import EPS

logger = logging.getLogger(__name__)
_UNFUSED = bool(os.environ.get('MM_RADAM_UNFUSED'))  # measurement knob: compose the update from Manifold calls


class RiemannianAdam(ManifoldOptimizer):
    graph_safe = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), nc=False, max_grad_norm=None, exact=False):
        if nc and betas[1] is not None:
            logger.warning('beta1=%.5f will be ignored because `nc` is True', betas[1])
        super().__init__(params, dict(lr=lr, betas=betas, nc=nc, max_grad_norm=max_grad_norm, exact=exact))
        self._tickets = {}  # per parameter: the device counter the fused kernel uses to advance state['step']

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tickets = {}

    @staticmethod
    def _counter(state, like):
        """state['step'] as a 0-dim fp64 device tensor (a reference checkpoint stores a Python int)."""
        t = state.get('step', 1)
        if not torch.is_tensor(t):
            t = torch.tensor(float(t), dtype=torch.float64, device=like.device)
            state['step'] = t
        return t

    def _update(self, group, p, state, manifold):
        if 'exp_avg' not in state:
            state['exp_avg'] = torch.zeros_like(p)      # first moment, a tangent vector at p
            state['exp_avg_sq'] = torch.zeros_like(p)   # ONE scalar per point, broadcast over it (radam.py:60)
        t = self._counter(state, p)
        m, v = state['exp_avg'], state['exp_avg_sq']
        beta1, beta2 = group['betas']
        fused = getattr(manifold, 'radam_step', None) if p.is_cuda and not _UNFUSED else None
        if fused is not None:
            # one launch: moments in place, the kernel advances the step counter (optim/radam.py:62-98)
            ticket = self._tickets.get(p)
            if ticket is None or ticket.device != p.device:
                ticket = torch.zeros(1, dtype=torch.int32, device=p.device)
                self._tickets[p] = ticket
            new_p = fused(p, p.grad, m, v, t, ticket, lr=group['lr'], betas=group['betas'], nc=group['nc'],
                          eps=EPS[p.dtype], max_grad_norm=group['max_grad_norm'], exact=group['exact'],
                          inplace=capturing(p))
            if new_p is not None:
                if new_p is not p:
                    assign(p, new_p)
                return
        if group['nc']:
            beta2 = 1.0 - 1.0 / t                        # AdamNc: the varying second-moment decay (radam.py:81-82)

        # the second moment sees the gradient norm BEFORE clipping (radam.py:72-74)
        rgrad = manifold.egrad2rgrad(p, p.grad)
        norm = manifold.norm(p, rgrad, keepdim=True)
        if group['max_grad_norm'] is not None:
            rgrad = rgrad * torch.clamp(group['max_grad_norm'] / norm, max=1.0)

        m.mul_(beta1).add_(rgrad, alpha=1 - beta1)
        v.mul_(beta2).add_(norm.pow(2) * (1 - beta2))
        bias = (1 - beta2**t)**0.5 / (1 - beta1**t)      # device scalars: no host round trip
        stride = (-group['lr'] * bias).to(p.dtype)
        direction = m / (v.sqrt() + EPS[p.dtype]) * stride
        move = manifold.exp if group['exact'] else manifold.retr
        new_p = move(p, direction)
        carried = manifold.transp(p, new_p, m)
        assign(p, new_p)
        assign(m, carried)
        t.add_(1)
