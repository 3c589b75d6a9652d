import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, contextlib, io
from oracle import loss as oloss, net as onet
from wseg_amd import synth, loss_aten
from wseg_amd.optim import PolyOptimizer
from wseg_amd.resnet38_contrast import Net
from wseg_amd.train import Trainer, second_view

# (A) loss code isolation: feed the ORACLE's CPU outputs to loss_aten on the GPU
n, size, seed = 2, 64, 31
sd = synth.procedural_state_dict(0)
img = synth.synthetic_images(n, size, seed); lab = synth.synthetic_labels(n, seed)
m1, m2 = synth.synthetic_dropout_masks(n, 2 * seed), synth.synthetic_dropout_masks(n, 2 * seed + 1)
ex = {}
ref = oloss.train_step(img, lab, dict(sd), m1, m2, 0.20, random.Random(5), ex)
bg_idx = torch.topk(torch.full((1, n * 256), 0.2), 32, dim=-1)[1][0]
o1 = tuple(t.detach().cuda() for t in ex["out1"]); o2 = tuple(t.detach().cuda() for t in ex["out2"])
got = loss_aten.step_loss(o1, o2, lab.cuda(), 0.20, random.Random(5), True, bg_idx)
for k in ref: print(f"A {k:18s} aten-gpu {float(got[k]):.7f}  oracle {float(ref[k]):.7f}  diff {float(got[k])-float(ref[k]):+.2e}")
got = loss_aten.step_loss(tuple(t.cpu() for t in o1), tuple(t.cpu() for t in o2), lab, 0.20, random.Random(5), True, bg_idx)
for k in ref: print(f"A' {k:18s} aten-cpu {float(got[k]):.7f}  oracle {float(ref[k]):.7f}  diff {float(got[k])-float(ref[k]):+.2e}")

# (B) lr = 0 multi-step: outputs must stay sane
dev = "cuda"
model = Net(precision="bf16")
with contextlib.redirect_stdout(io.StringIO()):
    groups = model.get_parameter_groups()
lr = 0.0
opt = PolyOptimizer([{'params': groups[0], 'lr': lr, 'weight_decay': 5e-4}, {'params': groups[1], 'lr': 2*lr, 'weight_decay': 0},
                     {'params': groups[2], 'lr': 10*lr, 'weight_decay': 5e-4}, {'params': groups[3], 'lr': 20*lr, 'weight_decay': 0}], lr=lr, weight_decay=5e-4, max_step=5000)
model.load_state_dict(synth.procedural_state_dict(0, device=dev)); model.cuda(); model.train()
tr = Trainer(model, opt, 0.20, random.Random(0), False)
img = synth.synthetic_images(2, 448, 0, dev); lab = synth.synthetic_labels(2, 0, dev)
for s in range(4):
    l = tr.step(img, lab)
    with torch.no_grad():
        model.set_dropout_masks(None)
        o = model(img)
    print("B", s, {k: round(float(v), 4) for k, v in l.items()}, "cam", float(o[0].abs().mean()), "fproj", float(o[2].abs().mean()), "rv", float(o[1].abs().mean()), flush=True)
