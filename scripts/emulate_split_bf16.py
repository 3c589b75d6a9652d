"""Would a split-bf16 conv mode (operands x = hi + lo in bf16, 3 MFMA products hi.hi + lo.hi + hi.lo, f32 accumulate — SURVEY.md §7 H2)
meet the parity bars of the exact-f32 mode (losses 1e-4, CAM arg-max 0 mismatches)?  Emulated on the CPU oracle (test infrastructure) by
replacing every conv2d of oracle/net.py with the three-product form in f32 arithmetic, then compared with the reference's own fixtures.
`python scripts/emulate_split_bf16.py [terms]`: terms 3 = hi/lo split (16-bit operands), 6 = hi/mid/lo split (24-bit operands, 6 products)."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as RF
from oracle import infer as oinfer, loss as oloss, net as onet
from wseg_amd import synth

TERMS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ROUND_RESIDUAL = len(sys.argv) > 2 and sys.argv[2] == "planes"      # also keep the residual stream (block outputs) as hi + lo planes (16-17 bits)


def split(x, n):
    parts, r = [], x
    for _ in range(n):
        p = r.bfloat16().float()
        parts.append(p)
        r = r - p
    return parts


class SplitF:
    def __getattr__(self, a):
        return getattr(RF, a)

    def conv2d(self, x, w, *a, **k):
        n = 2 if TERMS == 3 else 3
        xs, ws = split(x, n), split(w, n)
        out = None
        for i in range(n):
            for j in range(n):
                if i + j < n:                                   # 2-split: hh, hl, lh; 3-split: hh, hm, mh, hl, lh, mm
                    t = RF.conv2d(xs[i], ws[j], *a, **k)
                    out = t if out is None else out + t
        return out


G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sd = synth.procedural_state_dict(0)
onet.F = SplitF()
if ROUND_RESIDUAL:
    _res, _bot = onet.res_block, onet.bot_block

    def _r(x):
        hi = x.bfloat16().float()
        return hi + (x - hi).bfloat16().float()
    onet.res_block = lambda *a, **k: (lambda o: (_r(o[0]), o[1]))(_res(*a, **k))
    onet.bot_block = lambda *a, **k: (lambda o: (_r(o[0]), o[1]))(_bot(*a, **k))
for name in ("infer_1img", "infer_125x94", "infer_188x250"):
    g = np.load(os.path.join(G, name + ".npz"))
    H, W = int(g["H"]), int(g["W"])
    seed0 = int(g["seed0"]) if "seed0" in g.files else 40
    imgs = []
    for si, s in enumerate([0.5, 1.0, 1.5, 2.0]):
        im = synth.synthetic_images(1, (int(np.round(H * s)), int(np.round(W * s))), seed0 + si)
        imgs += [im, torch.flip(im, dims=[3])]
    norm, pred, _ = oinfer.infer_one(imgs, torch.from_numpy(g["label"]), sd, (H, W), 0.26)
    print(f"{TERMS} products{' + planes residual' if ROUND_RESIDUAL else ''}: {name}: arg-max mismatches {int((pred != g['pred']).sum())} of {pred.size}", flush=True)
g = np.load(os.path.join(G, "step_S128_N3.npz"))
n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
with torch.no_grad():
    out = oloss.train_step(synth.synthetic_images(n, size, seed), synth.synthetic_labels(n, seed), sd, synth.synthetic_dropout_masks(n, seed * 2),
                           synth.synthetic_dropout_masks(n, seed * 2 + 1), 0.20, random.Random(py_seed))
for k in ("loss", "loss_cls", "loss_er", "loss_ecr", "loss_nce", "loss_intra_nce", "loss_cross_nce", "loss_cross_nce2"):
    ref = float(g["s/" + k])
    print(f"{TERMS} products: step_S128_N3 {k}: {float(out[k]):.7f} vs {ref:.7f}  ({abs(float(out[k]) - ref) / max(1.0, abs(ref)):.2e} of max(1,|ref|))", flush=True)
