"""Fixed-cost probe of the 256-tile conv kernel: time vs K (IC sweep) at the 512-channel 56x56 geometry, both views batched.
A straight-line fit t = c + K*s gives the per-launch fixed cost c (prologue + epilogue + rounds tail) in K-tile units."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import _lib as L
from bench_conv import timeit

N, H, OC, k = 16, 56, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 3
H2 = 16
M = N * (H * H + H2 * H2)
for hint in (256, 224):
    pts = []
    for IC in (64, 128, 256, 512, 1024, 2048):
        x = torch.randn(M, IC, device="cuda").bfloat16()
        w = (torch.randn(OC, k * k, IC, device="cuda") * 0.02).bfloat16()
        y = torch.empty(M, OC, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: L.conv_igemm(x, w, y, N=N, IH=H, IW=H, IC=IC, OH=H, OW=H, OC=OC, seg2=(H2, H2, H2, H2), bm_hint=hint,
                                        KH=k, KW=k, stride=1, dil=1, pad=1), iters=20)
        kt = k * k * IC // 64
        pts.append((kt, t))
        print(f"bm{hint} OC={OC} IC={IC:5d} K-tiles {kt:4d}  {t*1e3:8.1f} us  {2.0*M*OC*IC*k*k/t/1e9:7.0f} TF/s", flush=True)
    (k1, t1), (k2, t2) = pts[2], pts[-1]
    s = (t2 - t1) / (k2 - k1)
    print(f"bm{hint}: slope {s*1e3:.3f} us per K-tile (asymptote {2.0*M*OC*64/s/1e9:.0f} TF/s), intercept {(t1 - s*k1)*1e3:.1f} us = {(t1 - s*k1)/s:.1f} K-tiles")
