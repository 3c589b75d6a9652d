#!/usr/bin/env python
"""One weight-gradient launch at a step shape, timed (probe builds: WSEG_WGRAD_DIAG=6 drops the LDS-DMA requests of the loop).   python scripts/bench_wgrad_one.py [512|256|d2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wseg_amd import _lib as L   # noqa: E402

SHAPES = {"512": (56, 512, 512, 3, 1), "256": (112, 256, 256, 3, 1), "d2": (56, 1024, 512, 3, 2), "1x1": (56, 2048, 1024, 1, 1)}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "512"
    H, IC, OC, k, d = SHAPES[which]
    dev, N = "cuda", 16
    H2 = H * 128 // 448
    M = N * (H * H + H2 * H2)
    x = torch.randn(M, IC, device=dev).relu().bfloat16()
    dy = torch.randn(M, OC, device=dev).bfloat16()
    dw = torch.zeros(OC, k * k, IC, device=dev, dtype=torch.float32)
    kw = dict(N=N, IH=H, IW=H, IC=IC, OH=H, OW=H, OC=OC, KH=k, KW=k, stride=1, dil=d, pad=d * (k // 2), seg2=(H2, H2, H2, H2))
    fn = lambda: L.conv_wgrad(x, dy, dw, **kw)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"wgrad {which}: {us:.1f} us per launch, {2.0 * M * OC * IC * k * k / us / 1e6:.0f} TF/s (WSEG_WGRAD_DIAG={os.environ.get('WSEG_WGRAD_DIAG', '0')})")


if __name__ == "__main__":
    main()
