"""Riemannian Adam with one second-moment scalar per point — counterpart of
graphembed/graphembed/optim/radam.py:12-98 (same constructor, param groups, state keys and update
order), built on the shared scaffold of `_common.py`.

Unlike the reference, the step counter lives in device memory and the bias corrections are computed
there, so a captured HIP graph of a training step replays correctly (`graph_safe`)."""
import logging
import os

import torch

from graphembed.optim._common import ManifoldOptimizer, assign, capturing, vector_layout
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
        elif t.device != like.device or t.dtype != torch.float64:  # e.g. restored by load_state_dict
            t = t.detach().to(device=like.device, dtype=torch.float64).reshape(())
            state['step'] = t
        return t

    def _moments(self, p):
        state = self.state[p]
        if 'exp_avg' not in state:
            state['exp_avg'] = torch.zeros_like(p)      # first moment, a tangent vector at p
            state['exp_avg_sq'] = torch.zeros_like(p)   # ONE scalar per point, broadcast over it (radam.py:60)
        return state['exp_avg'], state['exp_avg_sq'], self._counter(state, p)

    def _ticket(self, p):
        ticket = self._tickets.get(p)
        if ticket is None or ticket.device != p.device:
            ticket = torch.zeros(1, dtype=torch.int32, device=p.device)
            self._tickets[p] = ticket
        return ticket

    def _params_of(self, group):
        """All vector-space parameters of the group (Euclidean / Lorentz / sphere points, flat scales) take
        ONE launch per dtype (mm_vec_radam_step_multi); the rest goes through `_update`."""
        params = [p for p in group['params'] if p.grad is not None]
        if _UNFUSED or len(params) < 2:
            return params
        import ctypes
        from graphembed import _backend as B
        layouts = [vector_layout(p, self.manifold_of(p)) for p in params]
        rest = [p for p, lay in zip(params, layouts) if lay is None]
        by_dtype = {}
        for p, lay in zip(params, layouts):
            if lay is not None:
                by_dtype.setdefault((p.dtype, p.device), []).append((p, lay))
        lib = B.lib()
        most = lib.raw('mm_vec_rsgd_multi_max')()
        clip, betas = group['max_grad_norm'], group['betas']
        for (dtype, dev), items in by_dtype.items():
            if len(items) < 2:
                rest.extend(p for p, _ in items)
                continue
            for lo in range(0, len(items), most):
                chunk = items[lo:lo + most]
                k = len(chunk)
                moments = [self._moments(p) for p, _ in chunk]
                if not all(m.is_contiguous() and v.is_contiguous() and m.dtype == dtype and v.dtype == dtype
                           for m, v, _ in moments):
                    rest.extend(p for p, _ in chunk)
                    continue
                inplace = capturing(chunk[0][0])
                with B.on_device(dev):
                    xs = [p.detach() for p, _ in chunk]
                    outs = xs if inplace else [torch.empty_like(x) for x in xs]
                    lib.call('mm_vec_radam_step_multi', B.dtype_code(xs[0]), k,
                             (ctypes.c_int * k)(*[lay[0] for _, lay in chunk]), B.ptr_array(xs),
                             B.ptr_array([p.grad for p, _ in chunk]), B.ptr_array([m for m, _, _ in moments]),
                             B.ptr_array([v for _, v, _ in moments]), B.ptr_array([t for _, _, t in moments]),
                             B.ptr_array([self._ticket(p) for p, _ in chunk]),
                             (ctypes.c_int64 * k)(*[x.numel() // lay[1] for x, (_, lay) in zip(xs, chunk)]),
                             (ctypes.c_int * k)(*[lay[1] for _, lay in chunk]), float(group['lr']), float(betas[0]),
                             float(betas[1] if betas[1] is not None else 0.0), int(bool(group['nc'])),
                             float(EPS[dtype]), -1.0 if clip is None else float(clip), int(bool(group['exact'])),
                             B.ptr_array(outs), B.stream_of(xs[0]))
                if not inplace:
                    for (p, _), new in zip(chunk, outs):
                        assign(p, new)
        return rest

    def _update(self, group, p, state, manifold):
        m, v, t = self._moments(p)
        beta1, beta2 = group['betas']
        fused = getattr(manifold, 'radam_step', None) if p.is_cuda and not _UNFUSED else None
        if fused is not None:
            # one launch: moments in place, the kernel advances the step counter (optim/radam.py:62-98)
            ticket = self._ticket(p)
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
