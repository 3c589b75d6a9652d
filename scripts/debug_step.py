import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, contextlib, io
from oracle import loss as oloss
from wseg_amd import synth
from wseg_amd.optim import PolyOptimizer
from wseg_amd.resnet38_contrast import Net
from wseg_amd.train import Trainer

mode = sys.argv[1]
if mode == "parity":
    n, size, seed = 2, 64, 31
    sd = synth.procedural_state_dict(0)
    img = synth.synthetic_images(n, size, seed); lab = synth.synthetic_labels(n, seed)
    m1, m2 = synth.synthetic_dropout_masks(n, 2 * seed), synth.synthetic_dropout_masks(n, 2 * seed + 1)
    ex = {}
    ref = oloss.train_step(img, lab, dict(sd), m1, m2, 0.20, random.Random(5), ex)
    bg_idx = torch.topk(torch.full((1, n * 256), 0.2), 32, dim=-1)[1][0]
    model = Net(precision="fp32")
    with contextlib.redirect_stdout(io.StringIO()):
        groups = model.get_parameter_groups()
    opt = PolyOptimizer([{'params': groups[0], 'lr': 0.01, 'weight_decay': 5e-4}, {'params': groups[1], 'lr': 0.02, 'weight_decay': 0},
                         {'params': groups[2], 'lr': 0.1, 'weight_decay': 5e-4}, {'params': groups[3], 'lr': 0.2, 'weight_decay': 0}], lr=0.01, weight_decay=5e-4, max_step=100)
    model.load_state_dict(sd); model.cuda(); model.train(); model.set_dropout_masks([m1, m2])
    tr = Trainer(model, opt, 0.20, random.Random(5), rng_parity=True, bg_topk_idx=bg_idx)
    got = tr.step(img.cuda(), lab.cuda())
    for k in ref: print(f"{k:18s} hip {float(got[k]):.7f}  oracle {float(ref[k]):.7f}  diff {float(got[k])-float(ref[k]):+.2e}")
else:
    dev = "cuda"
    model = Net(precision="bf16")
    with contextlib.redirect_stdout(io.StringIO()):
        groups = model.get_parameter_groups()
    lr = float(sys.argv[2])
    opt = PolyOptimizer([{'params': groups[0], 'lr': lr, 'weight_decay': 5e-4}, {'params': groups[1], 'lr': 2*lr, 'weight_decay': 0},
                         {'params': groups[2], 'lr': 10*lr, 'weight_decay': 5e-4}, {'params': groups[3], 'lr': 20*lr, 'weight_decay': 0}], lr=lr, weight_decay=5e-4, max_step=5000)
    model.load_state_dict(synth.procedural_state_dict(0, device=dev)); model.cuda(); model.train()
    tr = Trainer(model, opt, 0.20, random.Random(0), False)
    img = synth.synthetic_images(4, 448, 0, dev); lab = synth.synthetic_labels(4, 0, dev)
    for s in range(8):
        l = tr.step(img, lab)
        eng = model._engine
        print(s, {k: round(float(v), 4) for k, v in l.items()}, "gnorm", float(eng.flat_g.norm()), "wnorm", float(eng.flat_w.norm()), flush=True)
