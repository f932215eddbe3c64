"""Riemannian Adam with one second-moment scalar per point — counterpart of
graphembed/graphembed/optim/radam.py:12-98 (same param groups, same update order).
All arithmetic goes through the Manifold API (HIP kernels) plus a few element-wise torch ops
on [n, point] tensors."""
import logging

import torch

from graphembed.modules import ManifoldParameter
from graphembed.optim.rsgd import _default_manifold
from graphembed.utils import EPS

logger = logging.getLogger(__name__)


class RiemannianAdam(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), nc=False, max_grad_norm=None, exact=False):
        if nc and betas[1] is not None:
            logger.warning('beta1=%.5f will be ignored because `nc` is True', betas[1])
        defaults = dict(lr=lr, betas=betas, nc=nc, max_grad_norm=max_grad_norm, exact=exact)
        super().__init__(params, defaults)

    def step(self, closure=None):
        loss = None
        if closure is not None:
            loss = closure()
        with torch.no_grad():
            for group in self.param_groups:
                self._step(group)
        return loss

    def _step(self, group):
        lr = group['lr']
        beta1, beta2 = group['betas']
        max_grad_norm = group['max_grad_norm']
        for x in group['params']:
            grad = x.grad
            if grad is None:
                continue
            state = self.state[x]
            if len(state) == 0:
                state['step'] = 1
                state['exp_avg'] = torch.zeros_like(x)
                state['exp_avg_sq'] = torch.zeros_like(x)   # one scalar per point, broadcast (radam.py:60)
            if isinstance(x, ManifoldParameter) and x.manifold is not None:
                manifold = x.manifold
            else:
                manifold = _default_manifold
            retr = manifold.exp if group['exact'] else manifold.retr

            grad = manifold.egrad2rgrad(x, grad)
            grad_norm = manifold.norm(x, grad, keepdim=True)        # norm BEFORE clipping (radam.py:72-74)
            if max_grad_norm is not None:
                grad = grad * torch.clamp(max_grad_norm / grad_norm, max=1.0)

            step = state['step']
            exp_avg, exp_avg_sq = state['exp_avg'], state['exp_avg_sq']
            if group['nc']:
                beta2 = 1 - 1 / step
            exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
            exp_avg_sq.mul_(beta2).add_(grad_norm.pow(2) * (1 - beta2))
            denom = exp_avg_sq.sqrt().add_(EPS[x.dtype])
            alpha = lr * (1 - beta2**step)**0.5 / (1 - beta1**step)
            direction = exp_avg / denom * (-alpha)
            new_x = retr(x, direction)
            exp_avg_new = manifold.transp(x, new_x, exp_avg)
            x.set_(new_x)
            exp_avg.set_(exp_avg_new)
            state['step'] += 1
