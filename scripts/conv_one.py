#!/usr/bin/env python
"""One conv layer shape launched repeatedly (for rocprofv3 counter passes over a single kernel):
    python scripts/conv_one.py [512|256|1x1|d2|d4] [reps] [hint]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wseg_amd import _lib as L   # noqa: E402

SHAPES = {"512": (56, 512, 512, 3, 1), "256": (112, 256, 256, 3, 1), "d2": (56, 1024, 512, 3, 2), "d4": (56, 1024, 2048, 3, 4), "1x1": (56, 2048, 1024, 1, 1)}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "512"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    hint = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    H, IC, OC, k, d = SHAPES[which]
    dev, N = "cuda", 16
    H2 = H * 128 // 448
    M = N * (H * H + H2 * H2)
    x = torch.randn(M, IC, device=dev).bfloat16()
    w = (torch.randn(OC, k * k, IC, device=dev) * 0.02).bfloat16()
    out2 = torch.empty(M, OC, device=dev, dtype=torch.bfloat16)
    sc, sh = torch.rand(OC, device=dev) + 0.5, torch.randn(OC, device=dev)
    geo = dict(N=N, IH=H, IW=H, IC=IC, OH=H, OW=H, OC=OC, KH=k, KW=k, stride=1, dil=d, pad=d * (k // 2), seg2=(H2, H2, H2, H2), bm_hint=hint)
    fn = lambda: L.conv_igemm(x, w, None, out2, scale=sc, shift=sh, **geo)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{which}: {us:.1f} us per launch, {2.0 * M * OC * IC * k * k / us / 1e6:.0f} TF/s")


if __name__ == "__main__":
    main()
