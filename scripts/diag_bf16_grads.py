import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import synth
from wseg_amd.resnet38_contrast import Net
from oracle import net as onet

torch.manual_seed(0)
sd0 = synth.procedural_state_dict(0)
n, size = 2, 64
x = synth.synthetic_images(n, size, 7)
masks = synth.synthetic_dropout_masks(n, 9)
sd = dict(sd0)
keys = onet.trainable_keys(sd)
for k in keys: sd[k] = sd[k].clone().requires_grad_(True)
ref = onet.net_forward(x, sd, masks)
g = torch.Generator().manual_seed(3)
ws = [torch.randn(o.shape, generator=g) / o.numel() ** 0.5 for o in ref]
which = sys.argv[1] if len(sys.argv) > 1 else "rand"
def fun(outs, ws):
    if which == "rand":
        return sum((o * w.to(o.device)).sum() for o, w in zip(outs, ws))
    # coherent functional: softplus of the GAP logits + mean of f_proj^2 + mean cam_rv
    return torch.nn.functional.softplus(outs[0].mean((2, 3))).sum() + (outs[2] ** 2).mean() + outs[1].mean() * 10 + outs[3].mean() * 10
fun(ref, ws).backward()
res = {}
for prec in ("fp32", "bf16"):
    m = Net(precision=prec); m.load_state_dict(sd0); m.cuda(); m.train()
    m.set_dropout_masks([masks])
    got = m(x.cuda())
    fun(got, ws).backward()
    params = dict(m.named_parameters())
    res[prec] = {k: params[k].grad.detach().cpu().clone() for k in keys}
    for nm, r, o in zip(["cam", "cam_rv", "f_proj", "rvd"], ref, got):
        print(prec, nm, "rel", float((o.detach().float().cpu() - r.detach()).abs().max() / r.detach().abs().max()))
for k in keys:
    rg = sd[k].grad
    e32 = float((res["fp32"][k] - rg).norm() / rg.norm())
    e16 = float((res["bf16"][k] - rg).norm() / rg.norm())
    cos = float(torch.nn.functional.cosine_similarity(res["bf16"][k].flatten(), rg.flatten(), dim=0))
    print(f"{k:32s} fp32 {e32:.2e}  bf16 {e16:.3f} cos {cos:.4f}  |g| {float(rg.norm()):.3e}")
