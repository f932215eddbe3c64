"""Scaffolding shared by the Riemannian optimizers of this package.

`ManifoldOptimizer` owns everything that is not the update rule itself: the lean `step` /
`zero_grad` (no profiler ranges or hook dispatch around updates that are one or two kernels long),
the lookup of a parameter's manifold (flat parameters — scales, curvatures — fall back to the
Euclidean rule, as graphembed/graphembed/optim/rsgd.py:7,56-59 does with `Euclidean(1)`), the
Riemannian gradient with its per-point norm clip, and assignment of new values in a way that
survives HIP-graph capture.  Subclasses implement `_update(group, p, state, manifold)`.
"""
import torch

from graphembed.modules import ManifoldParameter


class FlatRule:
    """Manifold-API subset for parameters that live in flat space, arithmetic in torch except for
    the fused momentum-free step, which reuses the Euclidean RSGD kernel on GPU tensors."""

    @staticmethod
    def egrad2rgrad(x, u):
        return u

    @staticmethod
    def norm(x, u, keepdim=False):
        return (u * u).sum(-1, keepdim=keepdim).clamp(min=1e-8).sqrt()

    @staticmethod
    def exp(x, u):
        return x + u

    retr = exp

    @staticmethod
    def transp(x, y, u):
        return u

    @staticmethod
    def rsgd_step(x, egrad, *, lr, max_grad_norm=None, exact=False, inplace=False):
        """One kernel instead of ~12 scalar framework kernels per parameter; None when the tensor
        is not eligible (CPU, wide last dimension), in which case the caller composes the step."""
        if not x.is_cuda:
            return None
        width = x.shape[-1] if x.ndim else 1
        if x.dtype not in (torch.float32, torch.float64):
            return None
        from graphembed import _backend as B
        if width > B.lib().raw("mm_vec_max_dim")():
            return None
        from graphembed import _backend as B
        xd = x.detach()
        inplace = inplace and xd.is_contiguous()
        xc = xd.reshape(-1, width).contiguous()
        gc = egrad.detach().reshape(-1, width).to(xc.dtype).contiguous()
        with B.on_device(xc.device):
            out = xc if inplace else torch.empty_like(xc)
            B.lib().call('mm_vec_rsgd_step', B.dtype_code(xc), B.EUCLIDEAN, B.ptr(xc), B.ptr(gc),
                         xc.shape[0], width, float(lr),
                         -1.0 if max_grad_norm is None else float(max_grad_norm), int(bool(exact)),
                         B.ptr(out), B.stream_of(xc))
        return x if inplace else out.reshape(x.shape)


    @staticmethod
    def rsgd_momentum_step(x, egrad, buf, *, lr, momentum, dampening, max_grad_norm=None, exact=False,
                           inplace=False):
        """The fused heavy-ball update of the Euclidean kernel for flat parameters (None when not eligible)."""
        if not x.is_cuda:
            return None
        from graphembed import _backend as B
        from graphembed.manifolds.vector import _vec_momentum
        width = x.shape[-1] if x.ndim else 1
        return _vec_momentum(B.EUCLIDEAN, width, x, egrad, buf, lr, momentum, dampening, max_grad_norm, exact,
                             inplace)

    @staticmethod
    def radam_step(x, egrad, exp_avg, exp_avg_sq, step, ticket, *, lr, betas, nc, eps, max_grad_norm=None,
                   exact=False, inplace=False):
        """The fused Adam update of the Euclidean kernel for flat parameters (None when not eligible)."""
        if not x.is_cuda:
            return None
        from graphembed import _backend as B
        from graphembed.manifolds.vector import _vec_radam
        width = x.shape[-1] if x.ndim else 1
        return _vec_radam(B.EUCLIDEAN, width, x, egrad, exp_avg, exp_avg_sq, step, ticket, lr, betas, nc, eps,
                          max_grad_norm, exact, inplace)


FLAT = FlatRule()


def capturing(t):
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


def assign(t, new):
    """`t.set_(new)` (what the reference does, rsgd.py:80-82) — except while a HIP graph is being
    captured, where the value is copied into t's own storage so that a replayed step keeps
    advancing the same memory."""
    if capturing(t):
        t.copy_(new)
    else:
        t.set_(new)


def vector_layout(p, manifold):
    """(kind, m) when `p` can be stepped by the multi-parameter vector kernels
    (mm_vec_rsgd_step_multi, mm_vec_radam_step_multi), else None."""
    import torch
    from graphembed import _backend as B
    from graphembed.manifolds.vector import VectorManifold
    if not p.is_cuda or p.dtype not in (torch.float32, torch.float64) or not p.is_contiguous():
        return None
    if p.grad.dtype != p.dtype or not p.grad.is_contiguous() or p.numel() == 0:
        return None
    if manifold is FLAT:
        kind, m = B.EUCLIDEAN, (p.shape[-1] if p.ndim else 1)
    elif isinstance(manifold, VectorManifold) and type(manifold).rsgd_step is VectorManifold.rsgd_step \
            and type(manifold).radam_step is VectorManifold.radam_step:
        kind, m = manifold._kind, manifold._m
    else:
        return None
    return (kind, m) if 1 <= m <= B.lib().raw("mm_vec_max_dim")() else None


class ManifoldOptimizer(torch.optim.Optimizer):

    def step(self, closure=None):
        hooked = self._has_step_hooks()
        if hooked:
            self._run_step_hooks(True, closure)
        loss = closure() if closure is not None else None
        with torch.no_grad():
            for group in self.param_groups:
                for p in self._params_of(group):
                    self._update(group, p, self.state[p], self.manifold_of(p))
        if hooked:
            self._run_step_hooks(False, closure)
        return loss

    def _params_of(self, group):
        """The parameters of `group` that `_update` has to visit (subclasses may serve some of them
        with a batched launch first)."""
        return [p for p in group['params'] if p.grad is not None]

    # torch.optim.Optimizer wraps `step` of every subclass in a profiler range plus pre/post hook
    # dispatch (~25 us of host time per call) unless it is marked as hooked already; the updates
    # here are a few microseconds of GPU work.  The wrapper is skipped, the hooks are not: `step`
    # dispatches registered step hooks (this optimizer's and the global ones) itself, in torch's
    # order, and pays for it only when there are any.
    step.hooked = True

    def _has_step_hooks(self):
        import torch.optim.optimizer as O
        return bool(getattr(self, '_optimizer_step_pre_hooks', None) or getattr(self, '_optimizer_step_post_hooks', None)
                    or getattr(O, '_global_optimizer_pre_hooks', None) or getattr(O, '_global_optimizer_post_hooks', None))

    def _run_step_hooks(self, pre, closure):
        """torch.optim.Optimizer.profile_hook_step's dispatch: global hooks first, then this optimizer's;
        hooks receive (optimizer, args, kwargs); a pre-hook may not rewrite the arguments here (step takes
        only the closure)."""
        import torch.optim.optimizer as O
        args, kwargs = ((closure, ) if closure is not None else ()), {}
        if pre:
            hooks = list(getattr(O, '_global_optimizer_pre_hooks', {}).values()) + \
                list(self._optimizer_step_pre_hooks.values())
        else:
            hooks = list(self._optimizer_step_post_hooks.values()) + \
                list(getattr(O, '_global_optimizer_post_hooks', {}).values())
        for hook in hooks:
            result = hook(self, args, kwargs)
            if pre and result is not None:
                raise RuntimeError(f'{type(self).__name__}: step pre-hooks that replace the arguments are not '
                                   'supported (the fused step takes only an optional closure)')

    def zero_grad(self, set_to_none=True):
        for group in self.param_groups:
            for p in group['params']:
                g = p.grad
                if g is None:
                    continue
                if set_to_none:
                    p.grad = None
                    continue
                if g.grad_fn is not None:
                    g.detach_()
                else:
                    g.requires_grad_(False)
                g.zero_()

    @staticmethod
    def manifold_of(p):
        if isinstance(p, ManifoldParameter) and p.manifold is not None:
            return p.manifold
        return FLAT

    @staticmethod
    def riemannian_gradient(manifold, p, max_grad_norm):
        """(clipped Riemannian gradient, its norm BEFORE clipping or None if not needed)."""
        rgrad = manifold.egrad2rgrad(p, p.grad)
        if max_grad_norm is None:
            return rgrad, None
        norm = manifold.norm(p, rgrad, keepdim=True)
        return rgrad * torch.clamp(max_grad_norm / norm, max=1.0), norm

    def _update(self, group, p, state, manifold):
        raise NotImplementedError
