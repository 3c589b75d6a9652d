"""Tile-kernel choice at INFERENCE geometry: one 375x500 image at scales 2.0 and 0.5 (+ flips) = N=2, segments 94x125 and 24x32
at stride 8 (and x2 / x4 for the earlier stages).  Prints fwd time per tile hint for the main layer shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import _lib as L
from bench_conv import timeit

SH = [("512->512 3x3", 8, 512, 512, 3, 1), ("512->1024 3x3 d2", 8, 512, 1024, 3, 2), ("1024->512 3x3 d2", 8, 1024, 512, 3, 2),
      ("1024->2048 3x3 d4", 8, 1024, 2048, 3, 4), ("2048->4096 1x1", 8, 2048, 4096, 1, 1), ("1024->2048 1x1", 8, 1024, 2048, 1, 1),
      ("256->256 3x3 /4", 4, 256, 256, 3, 1), ("128->128 3x3 /2", 2, 128, 128, 3, 1)]
N = 2
for pair in (((750, 1000), (188, 250)), ((563, 750), (375, 500))):
    for name, st, IC, OC, k, d in SH:
        (H, W), (H2, W2) = [((a + st - 1) // st, (b + st - 1) // st) for (a, b) in pair]
        M = N * (H * W + H2 * W2)
        x = torch.randn(M, IC, device="cuda").bfloat16()
        w = (torch.randn(OC, k * k, IC, device="cuda") * 0.02).bfloat16()
        y = torch.empty(M, OC, device="cuda", dtype=torch.bfloat16)
        line = f"{pair[0][0]}+{pair[1][0]} {name:20s} M={M:6d}"
        for hint in (0, 128, 256, 224, 259):
            if hint == 259 and OC % 128: continue
            t = timeit(lambda: L.conv_igemm(x, w, y, N=N, IH=H, IW=W, IC=IC, OH=H, OW=W, OC=OC, seg2=(H2, W2, H2, W2), bm_hint=hint,
                                            KH=k, KW=k, stride=1, dil=d, pad=d * (k // 2)), iters=10)
            line += f" | {hint}: {t*1e3:6.1f} us {2.0*M*OC*IC*k*k/t/1e9:5.0f} TF"
        print(line, flush=True)
