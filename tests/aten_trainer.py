"""Cross-check trainer (TEST INFRASTRUCTURE): the reference's loss body as device-side torch ops (tests/loss_aten.py) behind the
drop-in `Net.forward` 4-tuple — how a user of the reference's own training script would drive the drop-in module.  The product
`wseg_amd.train.Trainer` only runs the hand-written HIP loss kernels; this subclass exists so the GPU tests can hold those
kernels against an independent on-device formulation through the same network, all-reduce and optimizer code."""
from wseg_amd.train import Trainer, second_view

from . import loss_aten


class AtenTrainer(Trainer):
    def step(self, img1, label20):
        if not img1.is_cuda:
            raise RuntimeError("AtenTrainer.step needs GPU tensors")
        model = self.model
        model._engine.block_done_hook = None                # two separate backward passes: no per-block bucket hook
        img1 = img1.contiguous().float()
        img2 = second_view(img1)
        self.optimizer.zero_grad(flat=True)
        out1 = model(img1)
        out2 = model(img2)
        losses = loss_aten.step_loss(out1, out2, label20, self.bg_threshold, self.rng, self.rng_parity, self.bg_topk_idx)
        losses["loss"].backward()
        return self.finish_step(losses)
