#!/usr/bin/env python
"""bf16 step vs the oracle under the bf16 path's own decisions: per-column-group cosine of f9.weight's gradient (x_s | f8_3 | f8_4 columns) and of the
other PCM-branch keys.  (scripts/gpu.sh 600 'python scripts/diag_f9_bf16.py')"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pass
from tests import test_gpu_loss as T   # noqa: E402
from oracle import loss as oloss   # noqa: E402
from oracle import net as onet   # noqa: E402
from wseg_amd import synth   # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "step_S160_N2"
    prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    gd = os.path.join(ROOT, "tests", "golden")
    proc_sd = synth.procedural_state_dict(0)
    g = np.load(os.path.join(gd, name + ".npz"))
    n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
    model, opt, tr = T._trainer(proc_sd, prec, "hip", n, seed, py_seed)
    eng = model._engine
    eng.capture_ctx = True
    img, lab = synth.synthetic_images(n, size, seed), synth.synthetic_labels(n, seed)
    got = tr.step(img.cuda(), lab.cuda())
    S = eng.last_ctx
    gates = T.gates_from_ctx(S)
    for vi, vw in enumerate(S["views"]):
        h, w, off = vw["h"], vw["w"], vw["off"]
        gates[vi]["cam_d_norm"] = S["G"][off:off + n * h * w, :21].float().view(n, h, w, 21).permute(0, 3, 1, 2).contiguous().cpu()
    v1, v2 = eng.last_loss_views
    inject = dict(protos1=v1.protos.cpu(), protos2=v2.protos.cpu(), pseudo1=v1.y.cpu().long(), pseudo2=v2.y.cpu().long())
    sd = {k: v.clone() for k, v in proc_sd.items()}
    for k in onet.trainable_keys(sd):
        sd[k].requires_grad_(True)
    ref = oloss.train_step(img, lab, sd, synth.synthetic_dropout_masks(n, seed * 2), synth.synthetic_dropout_masks(n, seed * 2 + 1),
                           0.20, random.Random(py_seed), gates1=gates[0], gates2=gates[1], inject=inject)
    ref["loss"].backward()
    params = dict(model.named_parameters())

    def cos(a, b):
        a, b = a.reshape(-1).double(), b.reshape(-1).double()
        return float(a @ b / (a.norm() * b.norm() + 1e-300)), float(a.norm() / (b.norm() + 1e-300))
    for k in ("f9.weight", "f8_3.weight", "f8_4.weight", "fc8.weight", "fc_proj.weight"):
        if k not in params:
            continue
        a, b = params[k].grad.detach().cpu(), sd[k].grad
        print(k, "cos %.4f norm ratio %.4f" % cos(a, b), tuple(a.shape))
    a, b = params["f9.weight"].grad.detach().cpu()[:, :, 0, 0], sd["f9.weight"].grad[:, :, 0, 0]
    for nm, sl in (("x_s 0:3", slice(0, 3)), ("f8_3 3:67", slice(3, 67)), ("f8_4 67:195", slice(67, 195))):
        print("  f9 columns", nm, "cos %.4f norm ratio %.4f" % cos(a[:, sl], b[:, sl]), " |ref| %.3e" % float(b[:, sl].norm()))
    # rows (output channels) in quarters
    for q in range(4):
        print("  f9 rows", q * 48, "cos %.4f norm ratio %.4f" % cos(a[q * 48:(q + 1) * 48], b[q * 48:(q + 1) * 48]))


if __name__ == "__main__":
    main()
