"""One training step of contrast_train.py:128-399 on the MI355X path.

`Trainer.step(img1, label20)` = second view (bilinear 128x128, align_corners=True, :131-134) ->
two Net forwards -> loss -> backward -> gradient all-reduce (one process per GPU, RCCL) ->
PolyOptimizer step.  Returns the 8 logged scalars as device tensors (no host sync; the reference
syncs 8x per step with .item(), :401-408).
"""
import os
import random as _random

import torch
import torch.distributed as dist

from . import _lib as L
from .engine import DT_OF


def dist_state():
    """(world size, whether the data-parallel exchanges run).  They run whenever there is more than one rank;
    WSEG_FORCE_DIST=1 also runs them in an initialised one-rank group — the way the RCCL call path (candidate and
    hard-pixel all-gathers, bucketed gradient all-reduce) is exercised on a one-GPU box, where the results must equal the
    local path's."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1, False
    world = dist.get_world_size()
    return world, world > 1 or os.environ.get("WSEG_FORCE_DIST", "0") == "1"


def second_view(img1, size=128):
    """contrast_train.py:131-134 on the device (the reference does it on the host tensor)."""
    N, C, H, W = img1.shape
    out = torch.empty(N, C, size, size, device=img1.device, dtype=torch.float32)
    L.resize_planar_fwd(img1.contiguous(), out, N * C, H, W, size, size, True)
    return out


class Trainer:
    def __init__(self, model, optimizer, bg_threshold=0.20, rng=None, rng_parity=False, bg_topk_idx=None):
        self.model = model
        self.optimizer = optimizer
        self.bg_threshold = bg_threshold
        self.rng = rng if rng is not None else _random.Random()
        self.rng_parity = rng_parity
        self.bg_topk_idx = bg_topk_idx
        self.world, self.distributed = dist_state()
        self._pending = []
        if self.distributed and os.environ.get("WSEG_BUCKETS", "1") != "0":   # (one joint backward)
            # Gradient all-reduce overlapped with backward: the flat gradient buffer completes back to front, so each
            # bucket (b7 + heads, b5..b6, b4*, b3*: 154 / 143 / 109 / 13 MB) is reduced as soon as its last weight
            # gradient is enqueued — RCCL runs on its own stream behind those kernels while dgrad/wgrad continue.
            model._engine.block_done_hook = self._on_block_done

    def _on_block_done(self, name):
        eng = self.model._engine
        buckets = getattr(self, "_buckets", None)
        if buckets is None:
            buckets = self._buckets = eng.grad_buckets()
        if name in buckets:
            lo, hi = buckets[name]
            self._pending.append(dist.all_reduce(eng.flat_g[lo:hi], async_op=True))

    def step(self, img1, label20, next_img1=None):
        """next_img1 (optional): the images of the NEXT call.  The part of their forward pass that depends on no trainable weight (conv1a and
        the frozen b2 blocks) is then computed inside this step's loss phase, and the next call — which must pass that same tensor, unmodified —
        starts from it.  Results are identical with and without it."""
        if not img1.is_cuda:
            raise RuntimeError("Trainer.step needs GPU tensors (no CPU fallback)")
        from . import loss_hip
        img1 = img1.contiguous().float()
        pre, self._lookahead = getattr(self, "_lookahead", None), None
        # the prefix is reused only for the SAME image tensor AND the same frozen weights / BN buffers / precision it was computed from
        # (a load_state_dict or a BN-buffer edit between the two calls changes Engine.frozen_key: the stale activations are dropped)
        if (pre is not None and pre["img1"].data_ptr() == img1.data_ptr() and pre["img1"].shape == img1.shape and pre["version"] == img1._version
                and pre.get("fkey") == self.model._engine.frozen_key(img1.device, DT_OF[self.model.precision])):
            img2, prefix = pre["img2"], pre["prefix"]
        else:
            img2, prefix = second_view(img1), None
        la = None
        if next_img1 is not None and os.environ.get("WSEG_PREFETCH", "1") != "0":
            n1 = next_img1.contiguous().float()
            if n1.is_cuda and n1.shape == img1.shape:
                la = {"img1": n1, "version": n1._version}
        self.optimizer.zero_grad(flat=False)                # (the fused step clears the flat gradient buffer itself, off the critical path)
        losses = loss_hip.step(self.model, img1, img2, label20, self.bg_threshold, self.rng, self.rng_parity,
                               self.bg_topk_idx, zero_grads=True, prefix=prefix, lookahead=la)
        if la is not None:
            la["fkey"] = self.model._engine.frozen_key(img1.device, DT_OF[self.model.precision])
        self._lookahead = la
        return self.finish_step(losses)

    def finish_step(self, losses):
        """Gradient all-reduce (data parallel) + optimizer step; returns the logged scalars detached."""
        model, opt = self.model, self.optimizer
        if self.distributed:
            if self._pending:                               # bucketed all-reduces launched during backward
                for work in self._pending:
                    work.wait()
                self._pending = []
            else:
                dist.all_reduce(model._engine.flat_g)       # RCCL over xGMI; averaged by grad_scale below
            opt.wseg_grad_scale = 1.0 / self.world
        opt.step()
        return {k: v.detach() for k, v in losses.items()}
