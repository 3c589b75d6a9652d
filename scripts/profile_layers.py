"""Per-layer time / TFLOP/s of the conv launches inside real training steps (HIP events)."""
import sys, os, random, contextlib, io, collections
os.environ.setdefault("WSEG_BWD_PAIR", "0")      # per-layer table: data gradients and weight gradients as separate launches (the product pairs them)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import synth, _lib as L
from wseg_amd.optim import PolyOptimizer
from wseg_amd.resnet38_contrast import Net
from wseg_amd.train import Trainer
dev = "cuda"
model = Net(precision=sys.argv[1] if len(sys.argv) > 1 else "bf16")
with contextlib.redirect_stdout(io.StringIO()):
    groups = model.get_parameter_groups()
lr = 1e-5
opt = PolyOptimizer([{'params': groups[0], 'lr': lr, 'weight_decay': 5e-4}, {'params': groups[1], 'lr': 2*lr, 'weight_decay': 0},
                     {'params': groups[2], 'lr': 10*lr, 'weight_decay': 5e-4}, {'params': groups[3], 'lr': 20*lr, 'weight_decay': 0}], lr=lr, weight_decay=5e-4, max_step=5000)
model.load_state_dict(synth.procedural_state_dict(0, device=dev)); model.cuda(); model.train()
tr = Trainer(model, opt, 0.20, random.Random(0), False)
img = synth.synthetic_images(16, 448, 0, dev); lab = synth.synthetic_labels(16, 0, dev)
for _ in range(2): tr.step(img, lab)
L.PROFILE, L.PROFILE_WGRAD = [], []
steps = 3
for _ in range(steps): tr.step(img, lab)
torch.cuda.synchronize()
for nm, prof in (("igemm", L.PROFILE), ("wgrad", L.PROFILE_WGRAD)):
    agg = collections.OrderedDict()
    for (s, e, f, tag, *_rest) in prof:
        a = agg.setdefault(tag, [0, 0.0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e); a[2] += f
    tot = sum(a[1] for a in agg.values()) / steps
    print(f"== {nm}: {tot:.2f} ms/step, {sum(a[2] for a in agg.values())/steps/1e12:.2f} TFLOP/step")
    for tag, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {tag:40s} x{cnt//steps:3d}  {ms/steps:7.3f} ms/step  {fl/ms/1e9:7.1f} TF/s")
