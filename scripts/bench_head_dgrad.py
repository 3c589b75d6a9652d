#!/usr/bin/env python
"""The head's data gradient (fused head rows [M,192] -> dX [M,4096], BN-ReLU-backward mask): K = 192 is 3 K-tiles, the launch writes 444 MB and reads 444 MB of
mask — an HBM-roof launch.  Tile choices side by side.    python scripts/bench_head_dgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wseg_amd import _lib as L   # noqa: E402


def main():
    dev, N, H = "cuda", 16, 56
    H2 = 16
    M = N * (H * H + H2 * H2)
    K, OC = 192, 4096
    dy = torch.randn(M, K, device=dev).bfloat16()
    wt = (torch.randn(OC, 1, K, device=dev) * 0.05).bfloat16()
    mask = torch.randn(M, OC, device=dev).relu().bfloat16()
    out = torch.empty(M, OC, device=dev, dtype=torch.bfloat16)
    sc = torch.rand(OC, device=dev) + 0.5
    geo = dict(N=N, IH=H, IW=H, IC=K, OH=H, OW=H, OC=OC, KH=1, KW=1, mode=1, epi=1, scale=sc, mask=mask, seg2=(H2, H2, H2, H2))
    gb = (2 * M * OC * 2 + M * K * 2) / 1e9
    for hint in (0, 128, 64, 256, 224, 0):
        fn = lambda: L.conv_igemm(dy, wt, out, bm_hint=hint, **geo)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"bm_hint {hint:3d}: {us:7.1f} us  {gb / us * 1e3:6.2f} TB/s  {2.0 * M * OC * K / us / 1e6:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
