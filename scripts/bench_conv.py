"""Micro-benchmark of the conv kernels at the real layer shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import _lib as L

SHAPES = [  # name, N, H, IC, OC, k, stride, dil
    ("G10 512->512 3x3 56^2", 16, 56, 512, 512, 3, 1, 1),
    ("G19 1024->2048 3x3 d4", 16, 56, 1024, 2048, 3, 1, 4),
    ("G17 2048->4096 1x1", 16, 56, 2048, 4096, 1, 1, 1),
    ("G7 256->256 3x3 112^2", 16, 112, 256, 256, 3, 1, 1),
    ("G4 128->128 3x3 224^2", 16, 224, 128, 128, 3, 1, 1),
    ("G12 512->1024 3x3 d2", 16, 56, 512, 1024, 3, 1, 2),
    ("G10@128 512->512 16^2", 16, 16, 512, 512, 3, 1, 1),
]

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    tdt = torch.bfloat16 if which == "bf16" else torch.float32
    dev = "cuda"
    for name, N, H, IC, OC, k, s, d in SHAPES:
        if which == "f32" and H > 112: continue
        pad = d * (k // 2)
        OH = (H + 2 * pad - d * (k - 1) - 1) // s + 1
        x = torch.randn(N, H, H, IC, device=dev).to(tdt)
        wf = (torch.randn(OC, k * k, IC, device=dev) * 0.02).to(tdt)
        wt = (torch.randn(IC, k * k, OC, device=dev) * 0.02).to(tdt)
        y = torch.empty(N, OH, OH, OC, device=dev, dtype=tdt)
        dy = torch.randn(N, OH, OH, OC, device=dev).to(tdt)
        dx = torch.empty(N, H, H, IC, device=dev, dtype=tdt)
        dw = torch.zeros(OC, k * k, IC, device=dev)
        kw = dict(KH=k, KW=k, stride=s, dil=d, pad=pad)
        flop = 2.0 * N * OH * OH * OC * IC * k * k
        t_f = timeit(lambda: L.conv_igemm(x, wf, y, N=N, IH=H, IW=H, IC=IC, OH=OH, OW=OH, OC=OC, **kw))
        t_d = timeit(lambda: L.conv_igemm(dy, wt, dx, N=N, IH=OH, IW=OH, IC=OC, OH=H, OW=H, OC=IC, mode=1, **kw))
        t_w = timeit(lambda: L.conv_wgrad(x, dy, dw, N=N, IH=H, IW=H, IC=IC, OH=OH, OW=OH, OC=OC, **kw))
        print(f"{which} {name:28s} fwd {t_f:8.3f} ms {flop/t_f/1e9:7.1f} TF | dgrad {t_d:8.3f} ms {flop/t_d/1e9:7.1f} TF | wgrad {t_w:8.3f} ms {flop/t_w/1e9:7.1f} TF", flush=True)

if __name__ == "__main__":
    main()
