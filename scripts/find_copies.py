"""Lists where the device-to-device copies of one training step come from (torch profiler, Python stacks)."""
import sys, os, random, contextlib, io, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(img, lab)
    torch.cuda.synchronize()
agg = collections.Counter(); tim = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::cat", "aten::to", "aten::_to_copy"):
        st = [s for s in ev.stack if "wseg_amd" in s or "bench" in s][:2]
        key = (ev.name, " <- ".join(s.split("/")[-1] for s in st))
        agg[key] += 1; tim[key] += ev.device_time_total
for k, c in sorted(agg.items(), key=lambda kv: -tim[kv[0]]):
    print(f"{c:4d} x {tim[k]:9.1f} us  {k[0]:18s} {k[1]}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25))
