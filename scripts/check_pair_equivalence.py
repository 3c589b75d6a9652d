"""One B = 16 x 448 x 448 bf16 training step with fixed inputs, dropout masks and hard-pixel keys; writes the 8 scalars and the flat gradient buffer.
Run twice (WSEG_BWD_PAIR=0 / 1 — the switch is read once per process) and compare with `--compare a.pt b.pt`: the joint dgrad + wgrad grids must
give the gradients of the separate launches up to the order of the weight gradients' float atomics."""
import sys, os, random, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a["scalars"]:
        assert abs(a["scalars"][k] - b["scalars"][k]) <= 2e-6 * max(1.0, abs(a["scalars"][k])), (k, a["scalars"][k], b["scalars"][k])
    ga, gb = a["grad"], b["grad"]
    worst = 0.0
    for name, (off, cnt) in a["offsets"].items():
        x, y = ga[off:off + cnt], gb[off:off + cnt]
        rel = float((x - y).abs().max() / x.abs().max().clamp_min(1e-30))
        worst = max(worst, rel)
        assert rel <= 2e-3, (name, rel)
    print(f"pair vs separate launches, B = 16 x 448 x 448: 8 scalars equal to 2e-6, {len(a['offsets'])} gradient tensors equal to {worst:.1e} of their maximum (float atomics)")
    sys.exit(0)
from wseg_amd import synth
from wseg_amd.optim import PolyOptimizer
from wseg_amd.resnet38_contrast import Net
from wseg_amd.train import Trainer
dev = "cuda"
os.environ["WSEG_INTRA_KEY_SEED"] = "5"
torch.manual_seed(3)
model = Net(precision="bf16")
with contextlib.redirect_stdout(io.StringIO()):
    g = model.get_parameter_groups()
opt = PolyOptimizer([{'params': g[0], 'lr': 0.0, 'weight_decay': 5e-4}, {'params': g[1], 'lr': 0.0, 'weight_decay': 0},
                     {'params': g[2], 'lr': 0.0, 'weight_decay': 5e-4}, {'params': g[3], 'lr': 0.0, 'weight_decay': 0}], lr=0.0, weight_decay=5e-4, max_step=100)
model.load_state_dict(synth.procedural_state_dict(0, device=dev)); model.cuda(); model.train()
model.set_dropout_masks([synth.synthetic_dropout_masks(16, 10), synth.synthetic_dropout_masks(16, 11)])
tr = Trainer(model, opt, 0.20, random.Random(0), False)
losses = tr.step(synth.synthetic_images(16, 448, 7, dev), synth.synthetic_labels(16, 7, dev))
torch.cuda.synchronize()
eng = model._engine
torch.save({"scalars": {k: float(v) for k, v in losses.items()}, "grad": eng.flat_g.detach().cpu(), "offsets": dict(eng.offsets)}, sys.argv[1])
print("wrote", sys.argv[1], {k: round(float(v), 6) for k, v in losses.items()})
