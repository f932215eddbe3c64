"""Riemannian SGD over the Manifold API — counterpart of
graphembed/graphembed/optim/rsgd.py:10-82: same constructor and param-group keys, same update
order (egrad2rgrad -> per-point norm clip -> momentum with transport | plain exp/retr step)."""
from torch.optim.optimizer import required

from graphembed.optim._common import FLAT, ManifoldOptimizer, assign, capturing, vector_layout

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

    def _params_of(self, group):
        """Momentum-free groups first update every parameter that lives in a vector space (Euclidean /
        Lorentz / sphere points, flat scales) with ONE launch per dtype; what is left goes through
        `_update` one by one."""
        params = [p for p in group['params'] if p.grad is not None]
        if group['momentum'] != 0 or len(params) < 2:
            return params
        return _multi_vector_step(params, self.manifold_of, group)

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
        fused = getattr(manifold, 'rsgd_momentum_step', None) if p.is_cuda else None
        if fused is not None:  # one launch: the buffer is updated and transported in place
            new_p = fused(p, p.grad, state['momentum_buffer'], lr=lr, momentum=momentum,
                          dampening=group['dampening'], max_grad_norm=clip, exact=group['exact'],
                          inplace=capturing(p))
            if new_p is not None:
                if new_p is not p:
                    assign(p, new_p)
                return
        rgrad, _ = self.riemannian_gradient(manifold, p, clip)
        buf = state['momentum_buffer']
        buf.mul_(momentum).add_(rgrad, alpha=1 - group['dampening'])
        move = manifold.exp if group['exact'] else manifold.retr
        new_p = move(p, -lr * buf)
        carried = manifold.transp(p, new_p, buf)
        assign(p, new_p)
        assign(buf, carried)


def _multi_vector_step(params, manifold_of, group):
    import ctypes
    import torch
    from graphembed import _backend as B
    layouts = [vector_layout(p, manifold_of(p)) for p in params]
    rest = [p for p, lay in zip(params, layouts) if lay is None]
    by_dtype = {}
    for p, lay in zip(params, layouts):
        if lay is not None:
            by_dtype.setdefault((p.dtype, p.device), []).append((p, lay))
    lib = B.lib()
    most = lib.raw('mm_vec_rsgd_multi_max')()
    clip = group['max_grad_norm']
    for (dtype, dev), items in by_dtype.items():
        if len(items) < 2:
            rest.extend(p for p, _ in items)
            continue
        for lo in range(0, len(items), most):
            chunk = items[lo:lo + most]
            k = len(chunk)
            inplace = capturing(chunk[0][0])
            with B.on_device(dev):
                xs = [p.detach() for p, _ in chunk]
                outs = xs if inplace else [torch.empty_like(x) for x in xs]
                lib.call('mm_vec_rsgd_step_multi', B.dtype_code(xs[0]), k,
                         (ctypes.c_int * k)(*[lay[0] for _, lay in chunk]), B.ptr_array(xs),
                         B.ptr_array([p.grad for p, _ in chunk]),
                         (ctypes.c_int64 * k)(*[x.numel() // lay[1] for x, (_, lay) in zip(xs, chunk)]),
                         (ctypes.c_int * k)(*[lay[1] for _, lay in chunk]), float(group['lr']),
                         -1.0 if clip is None else float(clip), int(bool(group['exact'])),
                         B.ptr_array(outs), B.stream_of(xs[0]))
            if not inplace:
                for (p, _), new in zip(chunk, outs):
                    assign(p, new)
    return rest
