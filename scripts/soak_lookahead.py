"""Soak of the fused step's stream graph: the same 150 training steps (B = 16 x 448 x 448, bf16, four alternating synthetic batches) with and without the
one-batch lookahead; the loss of every step must be finite and the two trajectories must agree to the noise of the float atomics (a race between the
streams — a buffer reused while another stream still reads it — shows up as a diverging or non-finite trajectory), and the allocator must not grow."""
import sys, os, random, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import synth
from wseg_amd.optim import PolyOptimizer
from wseg_amd.resnet38_contrast import Net
from wseg_amd.train import Trainer
dev = "cuda"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
batches = [(synth.synthetic_images(16, 448, 100 + j, dev), synth.synthetic_labels(16, 100 + j, dev)) for j in range(4)]
os.environ["WSEG_INTRA_KEY_SEED"] = "3"
def run(look):
    torch.manual_seed(11)
    model = Net(precision="bf16")
    with contextlib.redirect_stdout(io.StringIO()):
        g = model.get_parameter_groups()
    lr = 1e-5
    opt = PolyOptimizer([{'params': g[0], 'lr': lr, 'weight_decay': 5e-4}, {'params': g[1], 'lr': 2 * lr, 'weight_decay': 0},
                         {'params': g[2], 'lr': 10 * lr, 'weight_decay': 5e-4}, {'params': g[3], 'lr': 20 * lr, 'weight_decay': 0}], lr=lr, weight_decay=5e-4, max_step=5000)
    model.load_state_dict(synth.procedural_state_dict(0, device=dev)); model.cuda(); model.train()
    tr = Trainer(model, opt, 0.20, random.Random(0), False)
    out, mem = [], []
    for i in range(steps):
        img, lab = batches[i % 4]
        l = tr.step(img, lab, next_img1=batches[(i + 1) % 4][0] if look else None)
        out.append(l["loss"])
        if i in (5, 20, steps // 2, steps - 1):
            torch.cuda.synchronize(); mem.append((round(torch.cuda.memory_allocated() / 2**30, 2), round(torch.cuda.memory_reserved() / 2**30, 2)))
    return torch.stack(out).cpu(), mem
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
if mode == "mem":                                            # one mode per process: allocator numbers of a fresh process
    look = sys.argv[3] == "1"
    _, m = run(look)
    print(f"lookahead={look}: (allocated, reserved) GiB after steps 5 / 20 / {steps // 2} / {steps}: {m}")
    assert m[-1][1] <= m[1][1] * 1.02 + 0.25, "reserved memory keeps growing"
    sys.exit(0)
a, _ = run(False)
a2, _ = run(False)
b, _ = run(True)
assert torch.isfinite(a).all() and torch.isfinite(b).all()
rel_same = ((a - a2).abs() / a.abs()).max().item()
rel = ((a - b).abs() / a.abs()).max().item()
print(f"{steps} steps: loss {a[0]:.5f} -> {a[-1]:.5f} (no lookahead), {a2[-1]:.5f} (no lookahead, second run), {b[-1]:.5f} (lookahead)")
print(f"max relative difference of a step's loss: two runs of the same schedule {rel_same:.2e}; lookahead vs none {rel:.2e}")
assert rel <= max(3 * rel_same, 1e-3), (rel, rel_same)
print("soak OK")
