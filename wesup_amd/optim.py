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
        self._all = len(params) == len(list(model.parameters()))
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

    @torch.no_grad()
    def step(self, closure=None):
        m = self.model
        self._sync_state_in()
        g = self.param_groups[0]
        lr, mu, wd = g['lr'], g['momentum'], g['weight_decay']
        if self._all and all(p.grad is not None and p.grad.data_ptr() == m._grad_views[n].data_ptr()
                             for n, p in m.named_parameters()):
            ops.sgd_step(m._flat, m._flat_grad, self._vflat, lr, mu, wd, self.grad_scale, self._first)
        else:                                   # frozen parameters or foreign grads: per-parameter launches
            for name, p in m.named_parameters():
                if not p.requires_grad or p.grad is None:
                    continue
                o, n = m._offs[name], (p.numel() + 63) // 64 * 64
                gv = m._flat_grad[o:o + n]
                if p.grad.data_ptr() != m._grad_views[name].data_ptr():
                    m._grad_views[name].copy_(p.grad)
                ops.sgd_step(m._flat[o:o + n], gv, self._vflat[o:o + n], lr, mu, wd, self.grad_scale, self._first)
        if mu != 0:
            for p, view in self._views.items():
                if p.requires_grad:
                    self.state[p]['momentum_buffer'] = view
        self._first = False
        return None
