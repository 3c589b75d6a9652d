"""PolyOptimizer — the reference's optimizer (tool/torchutils.py:11-33) on the fused HIP SGD kernel.

Same constructor and attributes (`global_step`, `max_step`, `param_groups[i]['lr']`).  The
reference's quirk is kept on purpose: `super().__init__(params, lr, weight_decay)`
hands weight_decay to torch.optim.SGD's *momentum* slot, so SGD runs with momentum 5e-4 and the
per-group weight decays; `momentum=0.9` is only the poly-decay exponent.

When every parameter with a gradient lives in a wseg_amd Engine's flat buffer (the normal case),
`step()` is one `wseg_sgd_step` launch over [params | grads | momentum]; parameters from anywhere
else are stepped by torch.optim.SGD itself.
"""
import torch

from . import _lib as L


class PolyOptimizer(torch.optim.SGD):
    def __init__(self, params, lr, weight_decay, max_step, momentum=0.9):
        super().__init__(params, lr, weight_decay)          # (sic) weight_decay -> SGD momentum
        self.global_step = 0
        self.max_step = max_step
        self.momentum = momentum
        self.__initial_lr = [group['lr'] for group in self.param_groups]
        self._flat_buf = None
        self._flat_first = True
        self.wseg_grad_scale = 1.0

    def _poly(self):
        if self.global_step < self.max_step:
            lr_mult = (1 - self.global_step / self.max_step) ** self.momentum
            for i in range(len(self.param_groups)):
                self.param_groups[i]['lr'] = self.__initial_lr[i] * lr_mult

    def _flat_plan(self):
        """Returns (engine, segments) when all params with grads are flat-backed, else None."""
        eng = None
        segs = []
        for g in self.param_groups:
            lo = hi = None
            for p in g['params']:
                if p.grad is None:
                    continue
                info = getattr(p, "_wseg_flat", None)
                if info is None:
                    return None
                e, off, n = info
                if eng is None:
                    eng = e
                if e is not eng or p.data_ptr() != e.flat_w.data_ptr() + 4 * off \
                        or p.grad.data_ptr() != e.flat_g.data_ptr() + 4 * off:
                    return None
                lo = off if lo is None else min(lo, off)
                hi = off + n if hi is None else max(hi, off + n)
            if lo is not None:
                segs.append((lo, hi, float(g['lr']), float(g['weight_decay']), float(g['momentum'])))
        if eng is None:
            return None
        # segments must tile the flat buffer exactly (every flat parameter has a gradient)
        segs.sort()
        pos = 0
        for s in segs:
            if s[0] != pos:
                return None
            pos = s[1]
        if pos != eng.flat_w.numel():
            return None
        return eng, segs

    @torch.no_grad()
    def step(self, closure=None):
        self._poly()
        plan = self._flat_plan()
        if plan is None:
            super().step(closure)
        else:
            eng, segs = plan
            if self._flat_buf is None or self._flat_buf.numel() != eng.flat_w.numel() or self._flat_buf.device != eng.flat_w.device:
                self._flat_buf = torch.empty_like(eng.flat_w)
                self._flat_first = True
            mirror = getattr(eng, "flat_wb", None)         # bf16 mode: the fused step refreshes the forward packs' mirror too
            if mirror is not None and (mirror.numel() != eng.flat_w.numel() or mirror.device != eng.flat_w.device):
                mirror = None
            L.sgd_step(eng.flat_w, eng.flat_g, self._flat_buf, [(s[0], s[1], s[2], s[3]) for s in segs],
                       segs[0][4], self.wseg_grad_scale, self._flat_first, mirror)
            self._flat_first = False
            eng.flat_w_version += 1
            if mirror is not None:
                eng.flat_wb_version = eng.flat_w_version
        self.global_step += 1

    def zero_grad(self, set_to_none=True, flat=True):
        """Flat-backed gradients are zeroed in place (one memset) instead of being dropped; flat=False leaves that memset to
        the caller (wseg_amd.train: the fused step does it on a side stream)."""
        done = set()
        for g in self.param_groups:
            for p in g['params']:
                info = getattr(p, "_wseg_flat", None)
                if info is not None and p.grad is not None and p.grad.data_ptr() == info[0].flat_g.data_ptr() + 4 * info[1]:
                    if id(info[0]) not in done:
                        if flat:
                            info[0].flat_g.zero_()
                        done.add(id(info[0]))
                elif p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()
