import sys, os, random, contextlib, io, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from wseg_amd import synth
from wseg_amd.optim import PolyOptimizer
from wseg_amd.resnet38_contrast import Net
from wseg_amd.train import Trainer
dev = "cuda"
model = Net(precision="bf16")
with contextlib.redirect_stdout(io.StringIO()):
    groups = model.get_parameter_groups()
lr = 1e-5
opt = PolyOptimizer([{'params': groups[0], 'lr': lr, 'weight_decay': 5e-4}, {'params': groups[1], 'lr': 2*lr, 'weight_decay': 0},
                     {'params': groups[2], 'lr': 10*lr, 'weight_decay': 5e-4}, {'params': groups[3], 'lr': 20*lr, 'weight_decay': 0}], lr=lr, weight_decay=5e-4, max_step=5000)
model.load_state_dict(synth.procedural_state_dict(0, device=dev)); model.cuda(); model.train()
tr = Trainer(model, opt, 0.20, random.Random(0), False)
img = synth.synthetic_images(16, 448, 0, dev); lab = synth.synthetic_labels(16, 0, dev)
for _ in range(3): tr.step(img, lab)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    tr.step(img, lab)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.2f} ms, until GPU done {1e3*(t2-t0):.2f} ms", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); tr.step(img, lab); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
