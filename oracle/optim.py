"""CPU restatement of PolyOptimizer (test infrastructure, see oracle/__init__.py).

tool/torchutils.py:11-33.  Quirk kept: `super().__init__(params, lr, weight_decay)`
passes weight_decay POSITIONALLY into torch.optim.SGD's `momentum` slot, so the effective SGD is
momentum = 5e-4 (the wt_dec value), dampening 0, nesterov off, and the per-group weight_decay comes
only from the group dicts (contrast_train.py:91-96); `momentum=0.9` is just the poly exponent.
"""
import torch


class PolySGD:
    def __init__(self, groups, lr, weight_decay, max_step, momentum=0.9):
        # groups: list of dict(params=[tensors], lr=..., weight_decay=...)
        self.groups = [dict(g) for g in groups]
        self.sgd_momentum = weight_decay           # the positional-argument quirk
        self.power = momentum
        self.global_step = 0
        self.max_step = max_step
        self.initial_lr = [g["lr"] for g in self.groups]
        self.bufs = {}

    def step(self, grads):
        """grads: dict id(param)->grad tensor or list parallel to params (None = skipped)."""
        if self.global_step < self.max_step:
            mult = (1 - self.global_step / self.max_step) ** self.power
            for g, lr0 in zip(self.groups, self.initial_lr):
                g["lr"] = lr0 * mult
        with torch.no_grad():
            for g in self.groups:
                for p in g["params"]:
                    d = grads.get(id(p))
                    if d is None:
                        continue
                    if g["weight_decay"] != 0:
                        d = d.add(p, alpha=g["weight_decay"])
                    buf = self.bufs.get(id(p))
                    if buf is None:
                        buf = self.bufs[id(p)] = d.clone()
                    else:
                        buf.mul_(self.sgd_momentum).add_(d)
                    p.add_(buf, alpha=-g["lr"])
        self.global_step += 1
