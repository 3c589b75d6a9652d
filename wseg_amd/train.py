"""One training step of contrast_train.py:128-399 on the MI355X path.

`Trainer.step(img1, label20)` = second view (bilinear 128x128, align_corners=True, :131-134) ->
two Net forwards -> loss -> backward -> gradient all-reduce (one process per GPU, RCCL) ->
PolyOptimizer step.  Returns the 8 logged scalars as device tensors (no host sync; the reference
syncs 8x per step with .item(), :401-408).
"""
import random as _random

import torch
import torch.distributed as dist

from . import _lib as L
from . import loss_aten


def second_view(img1, size=128):
    """contrast_train.py:131-134 on the device (the reference does it on the host tensor)."""
    N, C, H, W = img1.shape
    out = torch.empty(N, C, size, size, device=img1.device, dtype=torch.float32)
    L.resize_planar_fwd(img1.contiguous(), out, N * C, H, W, size, size, True)
    return out


class Trainer:
    def __init__(self, model, optimizer, bg_threshold=0.20, rng=None, rng_parity=False, loss_impl="hip",
                 bg_topk_idx=None):
        self.model = model
        self.optimizer = optimizer
        self.bg_threshold = bg_threshold
        self.rng = rng if rng is not None else _random.Random()
        self.rng_parity = rng_parity
        self.loss_impl = loss_impl
        self.bg_topk_idx = bg_topk_idx
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def step(self, img1, label20):
        if not img1.is_cuda:
            raise RuntimeError("Trainer.step needs GPU tensors (no CPU fallback)")
        model, opt = self.model, self.optimizer
        img1 = img1.contiguous().float()
        img2 = second_view(img1)
        opt.zero_grad()
        if self.loss_impl == "aten":
            out1 = model(img1)
            out2 = model(img2)
            losses = loss_aten.step_loss(out1, out2, label20, self.bg_threshold, self.rng, self.rng_parity,
                                         self.bg_topk_idx)
            losses["loss"].backward()
        else:
            from . import loss_hip
            losses = loss_hip.step(model, img1, img2, label20, self.bg_threshold, self.rng, self.rng_parity,
                                   self.bg_topk_idx)
        if self.world > 1:
            eng = model._engine
            dist.all_reduce(eng.flat_g)                     # RCCL over xGMI; averaged by grad_scale below
            opt.wseg_grad_scale = 1.0 / self.world
        opt.step()
        return {k: v.detach() for k, v in losses.items()}
