"""Micro-benchmark of the conv kernels at the real layer shapes, both views batched
(rows of the 448x448 view ++ rows of the 128x128 view, as the engine launches them).

    python scripts/bench_conv.py [bf16|f32] [hints, e.g. 128,256]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import _lib as L

SHAPES = [  # name, H (view 1; view 2 = H*128/448), IC, OC, k, stride, dil
    ("G10 512->512 3x3", 56, 512, 512, 3, 1, 1),
    ("G19 1024->2048 3x3 d4", 56, 1024, 2048, 3, 1, 4),
    ("G17 2048->4096 1x1", 56, 2048, 4096, 1, 1, 1),
    ("G17t 4096->2048 1x1", 56, 4096, 2048, 1, 1, 1),
    ("G7 256->256 3x3 112^2", 112, 256, 256, 3, 1, 1),
    ("G12 512->1024 3x3 d2", 56, 512, 1024, 3, 1, 2),
    ("G13 1024->512 3x3 d2", 56, 1024, 512, 3, 1, 2),
    ("G14 1024->2048 1x1", 56, 1024, 2048, 1, 1, 1),
    ("G18 2048->1024 1x1", 56, 2048, 1024, 1, 1, 1),
    ("G11 512->1024 1x1", 56, 512, 1024, 1, 1, 1),
    ("G9 256->512 3x3 s2", 112, 256, 512, 3, 2, 1),
    ("G4 128->128 3x3 224^2", 224, 128, 128, 3, 1, 1),
    ("G3 64->128 3x3 s2 448^2", 448, 64, 128, 3, 2, 1),
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
    hints = [int(h) for h in (sys.argv[2] if len(sys.argv) > 2 else "128,256").split(",")]
    tdt = torch.bfloat16 if which == "bf16" else torch.float32
    dev = "cuda"
    N = 16
    only = [a_[7:] for a_ in sys.argv if a_.startswith("--only=")]
    for name, H, IC, OC, k, s, d in SHAPES:
        if only and not any(o in name for o in only):
            continue
        pad = d * (k // 2)
        osz = lambda h: (h + 2 * pad - d * (k - 1) - 1) // s + 1
        H2 = H * 128 // 448
        OH, OH2 = osz(H), osz(H2)
        Mi, Mo = N * (H * H + H2 * H2), N * (OH * OH + OH2 * OH2)
        x = torch.randn(Mi, IC, device=dev).to(tdt)
        wf = (torch.randn(OC, k * k, IC, device=dev) * 0.02).to(tdt)
        wt = (torch.randn(IC, k * k, OC, device=dev) * 0.02).to(tdt)
        y = torch.empty(Mo, OC, device=dev, dtype=tdt)
        dy = torch.randn(Mo, OC, device=dev).to(tdt)
        dx = torch.empty(Mi, IC, device=dev, dtype=tdt)
        kw = dict(KH=k, KW=k, stride=s, dil=d, pad=pad)
        flop = 2.0 * Mo * OC * IC * k * k
        line = f"{which} {name:24s} M={Mo:6d}"
        ref_y = ref_dx = None
        for h in hints:
            t_f = timeit(lambda: L.conv_igemm(x, wf, y, N=N, IH=H, IW=H, IC=IC, OH=OH, OW=OH, OC=OC, seg2=(H2, H2, OH2, OH2), bm_hint=h, **kw))
            if ref_y is None: ref_y = y.float().clone()
            else: assert float((y.float() - ref_y).abs().max()) <= 0.05 * float(ref_y.abs().max()), "fwd mismatch between tile kernels"
            t_d = timeit(lambda: L.conv_igemm(dy, wt, dx, N=N, IH=OH, IW=OH, IC=OC, OH=H, OW=H, OC=IC, mode=1, seg2=(OH2, OH2, H2, H2), bm_hint=h, **kw))
            if ref_dx is None: ref_dx = dx.float().clone()
            else: assert float((dx.float() - ref_dx).abs().max()) <= 0.05 * float(ref_dx.abs().max()), "dgrad mismatch between tile kernels"
            line += f" | bm{h}: fwd {t_f:6.3f} ms {flop/t_f/1e9:6.0f} TF  dgrad {t_d:6.3f} ms {flop/t_d/1e9:6.0f} TF"
        if "--wgrad" in sys.argv:
            dw = torch.zeros(OC, k * k, IC, device=dev)
            t_w = timeit(lambda: L.conv_wgrad(x, dy, dw, N=N, IH=H, IW=H, IC=IC, OH=OH, OW=OH, OC=OC, seg2=(H2, H2, OH2, OH2), **kw))
            line += f" | wgrad {t_w:6.3f} ms {flop/t_w/1e9:6.0f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
