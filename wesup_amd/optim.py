"""Fused SGD over the model's flat parameter buffer (replaces torch.optim.SGD.step,
models/wesup.py:445-451; semantics of SURVEY.md Appendix A: g += wd*p; buf = mu*buf + g
(first step buf = g); p -= lr*buf).  It IS a torch.optim.SGD (same param_groups and
state_dict layout, momentum buffers under 'momentum_buffer'), only ``step`` is replaced by one
launch of the ``wesup_sgd_step`` kernel; ``grad_scale`` folds the 1/world_size of data-parallel
gradient averaging into the same pass."""
import torch

from . import ops


class FusedSGD(torch.optim.SGD):
    def __init__(self, model, lr=5e-5, momentum=0.9, weight_decay=0.0, grad_scale=1.0):
        model._ensure_engine()
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, lr=lr, momentum=momentum or 0.0, weight_decay=weight_decay or 0.0)
        self.grad_scale = grad_scale
        self._vflat = torch.zeros_like(model._flat)
        self._first = True
        self._views = {}
        for name, p in model.named_parameters():
            o, n = model._offs[name], p.numel()
            self._views[p] = self._vflat[o:o + n].view(p.shape)

    def _sync_state_in(self):
        """Adopt momentum buffers that load_state_dict() put into self.state."""
        for p, view in self._views.items():
            st = self.state.get(p)
            if st and st.get('momentum_buffer') is not None and st['momentum_buffer'].data_ptr() != view.data_ptr():
                view.copy_(st['momentum_buffer'])
                st['momentum_buffer'] = view
                self._first = False

    def _trainable_ranges(self, with_grad):
        """Contiguous [lo, hi) element ranges of the flat buffers that hold parameters that are stepped (padding
        included, adjacent parameters merged): one range = the whole buffer when nothing is frozen, the tail behind the
        backbone with freeze_backbone (models/wesup.py:427-429,447).  with_grad: one flag per parameter -- like
        torch.optim.SGD, a parameter whose ``grad`` is None (no backward since zero_grad(): a raised or partial backward,
        step() called twice) is skipped, not updated with whatever the flat gradient buffer still holds."""
        m = self.model
        ranges = []
        for (name, p), has in zip(m._named, with_grad):
            if not has:
                continue
            lo, hi = m._offs[name], m._offs[name] + (p.numel() + 63) // 64 * 64
            if ranges and ranges[-1][1] == lo:
                ranges[-1][1] = hi
            else:
                ranges.append([lo, hi])
        return ranges

    def _prepare(self):
        m = self.model
        self._sync_state_in()
        m._ensure_engine()
        for name, p in m._named:                      # a gradient that is not the flat view (set by hand): adopt it
            if p.requires_grad and p.grad is not None and p.grad.data_ptr() != m._grad_views[name].data_ptr():
                m._grad_views[name].copy_(p.grad)
        sig = tuple(p.requires_grad and p.grad is not None for _, p in m._named)
        if getattr(self, '_ranges_sig', None) != sig:
            self._ranges, self._ranges_sig = self._trainable_ranges(sig), sig

    def _launch(self, ranges):
        m = self.model
        g = self.param_groups[0]
        lr, mu, wd = g['lr'], g['momentum'], g['weight_decay']
        for lo, hi in ranges:                         # one launch per contiguous trainable range
            ops.sgd_step(m._flat[lo:hi], m._flat_grad[lo:hi], self._vflat[lo:hi], lr, mu, wd, self.grad_scale, self._first)

    def _finish(self):
        if self.param_groups[0]['momentum'] != 0:
            for p, view in self._views.items():
                if p.requires_grad and p.grad is not None:
                    self.state[p]['momentum_buffer'] = view
        self._first = False

    @torch.no_grad()
    def step(self, closure=None):
        self._prepare()
        self._launch(self._ranges)
        self._finish()
        return None

    # The step in two launches (the step runner, one rank): every parameter but those named in ``late`` as soon as their gradients
    # are complete -- beside the end of the backward pass, on the stream the caller has made current -- and the rest behind it.
    # The update is elementwise: the two parts together are step(), bit for bit.
    @torch.no_grad()
    def step_early(self, late):
        m = self.model
        self._prepare()
        cut = sorted((m._offs[n], m._offs[n] + (dict(m._named)[n].numel() + 63) // 64 * 64) for n in late)
        early, held = [], []
        for lo, hi in self._ranges:
            pos = lo
            for a, b in cut:
                a, b = max(a, lo), min(b, hi)
                if a >= b:
                    continue
                if a > pos:
                    early.append([pos, a])
                if held and held[-1][1] == a:             # (weight and bias of a layer are neighbours: one launch)
                    held[-1][1] = b
                else:
                    held.append([a, b])
                pos = b
            if pos < hi:
                early.append([pos, hi])
        self._late = held
        self._launch(early)

    @torch.no_grad()
    def step_late(self):
        self._launch(self._late)
        self._late = None
        self._finish()
