"""Losses on vectors of *squared* distances used by the path's configs —
counterparts of graphembed/graphembed/objectives.py:16-45 (Stress, Quotient).
PearsonRLoss runs on the differentiable pdist path; the KL / curvature losses (inference models, graph
sampling) are outside the accelerated path."""
import abc

import torch


class ObjectiveFunction:

    @abc.abstractmethod
    def __call__(self, gdists, mdists, *, epoch, alpha):
        pass


class QuotientLoss(ObjectiveFunction):
    """|m/(a g) - 1| (+ |a g/(m + 1/(epoch+1)) - 1|), summed (objectives.py:16-36)."""

    def __init__(self, inc_l1=True, inc_l2=True):
        if not inc_l1 and not inc_l2:
            raise ValueError('At least one of the terms must be included.')
        self.inc_l1 = inc_l1
        self.inc_l2 = inc_l2

    def __call__(self, gdists, mdists, *, epoch, alpha):
        gdists = gdists * alpha
        loss = 0
        if self.inc_l1:
            loss = loss + (mdists / gdists - 1.0).abs().sum()
        if self.inc_l2:
            loss = loss + (gdists / (mdists + 1.0 / (epoch + 1)) - 1.0).abs().sum()
        return loss

    _dyn = None
    _dyn_host = None     # pinned staging buffer of the device copy
    _dyn_value = None    # (alpha, eps) last written to the device copy

    def on_device(self, device):
        """Keep {alpha, eps = 1/(epoch+1)} in device memory from now on: the fused kernels then read them
        there, so a captured HIP graph of a training step (graphembed.graphed) follows the loss's per-epoch
        schedule — call `set_epoch(epoch, alpha)` before each replay — instead of being re-recorded.
        Returns the fp64 tensor [alpha, eps]."""
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if self._dyn is None or self._dyn.device != device:
            self._dyn = torch.tensor([1.0, 1.0], dtype=torch.float64, device=device)
            self._dyn_host = torch.empty(2, dtype=torch.float64,
                                         pin_memory=device.type == 'cuda' and torch.cuda.is_available())
            self._dyn_value = (1.0, 1.0)
        return self._dyn

    def set_epoch(self, epoch, alpha, force=False):
        """Writes the schedule of `epoch` into the device-resident parameters (after `on_device`): nothing
        when the values are the ones already there, else an asynchronous copy from a pinned staging buffer
        on the CURRENT stream (the staging buffer is rewritten only after the previous copy has been consumed).
        Stream order makes the new values visible to kernels launched later on that stream; a consumer on
        ANOTHER stream (a graph replayed from a side stream) must wait for the copy — `fused_spec` does so
        for the calling stream, and `wait_schedule()` does it explicitly.  `force=True` rewrites the device
        copy even if the host believes it is current (after the tensor was restored / overwritten externally)."""
        value = (float(alpha), 1.0 / (epoch + 1))
        if value == self._dyn_value and not force:
            return
        if self._dyn.is_cuda:
            if getattr(self, '_dyn_event', None) is not None:
                self._dyn_event.synchronize()   # the previous copy has left the staging buffer (rarely waits)
            self._dyn_host[0], self._dyn_host[1] = value
            self._dyn.copy_(self._dyn_host, non_blocking=True)
            self._dyn_event = torch.cuda.Event()
            self._dyn_event.record()
            self._dyn_stream = torch.cuda.current_stream(self._dyn.device).cuda_stream
        else:
            self._dyn[0], self._dyn[1] = value
        self._dyn_value = value

    def wait_schedule(self):
        """Makes the current stream wait for the last `set_epoch` copy if that was issued on another stream."""
        ev = getattr(self, '_dyn_event', None)
        if ev is None or self._dyn is None or not self._dyn.is_cuda:
            return
        cur = torch.cuda.current_stream(self._dyn.device)
        if cur.cuda_stream != getattr(self, '_dyn_stream', None) and not ev.query():
            cur.wait_event(ev)

    def fused_spec(self, *, epoch, alpha):
        """(kind, alpha, eps, terms[, device {alpha, eps}]) for the fused loss+gradient kernels
        (mm_*_pdist_loss).  After `on_device` the values of this call are also written to the device
        copy — except while a graph is being recorded, where the device copy is what counts."""
        eps = 1.0 / (epoch + 1)
        terms = int(self.inc_l1) | (int(self.inc_l2) << 1)
        if self._dyn is None:
            return ('quotient', float(alpha), eps, terms)
        if not (self._dyn.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.set_epoch(epoch, alpha)
            self.wait_schedule()
        return ('quotient', float(alpha), eps, terms, self._dyn)

    def __str__(self):
        return 'quotient_loss'


class StressLoss(ObjectiveFunction):
    """sum (m - g)^2 (objectives.py:39-45)."""

    def __call__(self, gdists, mdists, *, epoch=None, alpha=None):
        return torch.pow(mdists - gdists, 2).sum()

    def fused_spec(self, *, epoch=None, alpha=None):
        return ('stress', 1.0, 0.0, 0)

    def __str__(self):
        return 'stress_loss'


class PearsonRLoss(ObjectiveFunction):
    """Negative Pearson correlation of the two pair vectors (objectives.py:109-117).  No fused kernel: it
    runs on the differentiable pdist path (`compute_dists`)."""

    def __call__(self, x, y, **kwargs):
        dx, dy = x - x.mean(), y - y.mean()
        return -(dx * dy).sum() / (dx.norm() * dy.norm())

    def __str__(self):
        return 'pearson_r_loss'


class Sum(ObjectiveFunction):

    def __init__(self, *fns):
        self.fns = fns

    def __call__(self, *args, **kwargs):
        return sum(fn(*args, **kwargs) for fn in self.fns)

    def __str__(self):
        return '__'.join(str(f) for f in self.fns)
