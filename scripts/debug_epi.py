import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import _lib as L
dev = "cuda"
for tdt in (torch.float32, torch.bfloat16):
    N, H, W, IC, OC, k = 2, 12, 10, 64, 128, 3
    xg = torch.randn(N, H, W, IC, device=dev).to(tdt)
    wf = (torch.randn(OC, k * k, IC, device=dev) * 0.05).to(tdt)
    out = torch.empty(N, H, W, OC, device=dev, dtype=tdt)
    for name, kw in [("plain", {}), ("epi2", dict(epi=2)), ("epi2-ld", dict(epi=2, ld_out=256, wide=True)),
                     ("epi0-scale", dict(scale=torch.ones(OC, device=dev), shift=torch.zeros(OC, device=dev), out2=True)),
                     ("epi1-mask", dict(epi=1, mask=torch.randn(N, H, W, OC, device=dev).to(tdt)))]:
        print(tdt, name, flush=True)
        kw = dict(kw)
        o = out
        if kw.pop("wide", False):
            wide = torch.zeros(N, H, W, 256, device=dev, dtype=tdt); o = wide[..., 64:]
        o2 = torch.empty_like(out) if kw.pop("out2", False) else None
        L.conv_igemm(xg, wf, o, o2, N=N, IH=H, IW=W, IC=IC, OH=H, OW=W, OC=OC, KH=k, KW=k, pad=1, **kw)
        torch.cuda.synchronize()
        print("  ok", float(o.float().abs().mean()), flush=True)
