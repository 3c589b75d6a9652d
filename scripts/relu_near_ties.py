"""Why a fixture seed can disagree between two correct f32 implementations: ReLU pre-activations within f32 summation
noise of zero that carry a large gradient.  Runs the CPU oracle step (oracle/, test infrastructure) at the edge-label
configuration of oracle/make_goldens.py and lists pre-activations below 2e-6 of their layer's maximum whose output gradient
is above 10 % of the layer's largest.  The all-twenty-classes image produces dozens per step at 64x64; whether one flips
depends on the summation order of the conv that feeds it.  With seed 13 the b5_2 element (x = +2.9e-5 against a layer
maximum of 90, 55 % of the largest gradient) flips on the GPU and every earlier layer's gradient moves by ~2 % while all
loss scalars and all later layers agree to 1e-6; with seed 16 — the committed fixture — a smaller one flips before b4
(0.4 %).  Hence the 5e-2 backbone bar of the edge fixture in tests/test_gpu_loss.py.

    python scripts/relu_near_ties.py [seed ...]
"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as RF
from oracle import loss as oloss, net as onet
from wseg_amd import synth


class _Recorder:
    def __init__(self): self.calls = []
    def __getattr__(self, a): return getattr(RF, a)
    def relu(self, x, *a, **k):
        y = RF.relu(x)
        if x.requires_grad:
            y.retain_grad(); self.calls.append((x.detach(), y))
        return y


def main():
    n, size = 3, 64
    for seed in [int(s) for s in sys.argv[1:]] or [13, 16]:
        lab = synth.synthetic_labels(n, seed).clone(); lab[0] = 0; lab[1] = 1
        sd = dict(synth.procedural_state_dict(0))
        for k in onet.trainable_keys(sd): sd[k] = sd[k].clone().requires_grad_(True)
        rec = onet.F = _Recorder()
        ref = oloss.train_step(synth.synthetic_images(n, size, seed), lab, sd, synth.synthetic_dropout_masks(n, 2 * seed),
                               synth.synthetic_dropout_masks(n, 2 * seed + 1), 0.20, random.Random(5))
        ref["loss"].backward()
        onet.F = RF
        worst = 0.0
        for i, (x, y) in enumerate(rec.calls):
            if y.grad is None: continue
            near = x.abs() < 2e-6 * x.abs().max()
            for j in near.nonzero():
                j = tuple(int(v) for v in j)
                rel = abs(float(y.grad[j])) / float(y.grad.abs().max())
                worst = max(worst, rel)
                if rel > 0.1:
                    print(f"seed {seed}: relu#{i} {tuple(x.shape)} at {j}: x={float(x[j]):+.2e} (layer max {float(x.abs().max()):.1e}) "
                          f"carries {100 * rel:.0f} % of the layer's largest gradient")
        print(f"seed {seed}: largest near-tie gradient share {100 * worst:.0f} %")


if __name__ == "__main__":
    main()
