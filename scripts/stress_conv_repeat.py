#!/usr/bin/env python
"""Run-to-run determinism of the 256-tile conv: the same launch repeated, every output compared bit for bit with the first one (a data race inside the
pipeline would show as rare differing tiles).  Cases: the two-source data gradient with a mask (stand-alone and inside the paired grid), a 3x3 forward with
two row segments, a 3x3 data gradient.    python scripts/stress_conv_repeat.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wseg_amd import _lib as L   # noqa: E402


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev, tdt = "cuda", torch.bfloat16
    # ---- two-source data gradient (tests/test_gpu_conv.py::test_conv_bwd_pair_two_sources)
    N, H, W, C1, C2, OCd = 2, 96, 100, 512, 128, 256
    M = N * H * W
    D, du1, t = rnd((M, C1), 1).to(dev, tdt), rnd((M, C2), 2).to(dev, tdt), rnd((M, OCd), 3).to(dev, tdt)
    mask = (rnd((M, OCd), 4) > 0).to(dev, tdt)
    wcat = rnd((OCd, 1, C1 + C2), 5, (1.0 / (C1 + C2)) ** 0.5).to(dev, tdt)
    scale = (rnd((OCd,), 6) + 1.5).to(dev)
    kw = dict(N=N, IH=H, IW=W, IC=C1, OH=H, OW=W, OC=OCd, KH=1, KW=1, mode=1, in2=du1, IC2=C2, epi=1, scale=scale, mask=mask)
    wkw = dict(N=N, IH=H, IW=W, IC=OCd, OH=H, OW=W, OC=C1, KH=1, KW=1)
    ref = torch.empty(M, OCd, device=dev, dtype=tdt)
    L.conv_igemm(D, wcat, ref, **kw)
    bad_a = bad_b = 0
    out = torch.empty_like(ref)
    dw = torch.zeros(C1, 1, OCd, device=dev, dtype=torch.float32)
    for _ in range(reps):
        out.fill_(float("nan")); L.conv_igemm(D, wcat, out, **kw)
        bad_a += int(not torch.equal(out, ref))
        out.fill_(float("nan")); L.conv_igemm(D, wcat, out, pair_wgrad=(t, D, dw, wkw), **kw)
        bad_b += int(not torch.equal(out, ref))
    print(f"two-source dgrad: {bad_a} / {reps} stand-alone and {bad_b} / {reps} paired launches differ from the first")
    # ---- 3x3 forward and data gradient, two row segments, 512 channels (224-row tiles)
    N, H, C = 4, 56, 512
    H2 = 16
    M = N * (H * H + H2 * H2)
    x = rnd((M, C), 11).to(dev, tdt)
    w = rnd((C, 9, C), 12, 0.02).to(dev, tdt)
    geo = dict(N=N, IH=H, IW=H, IC=C, OH=H, OW=H, OC=C, KH=3, KW=3, stride=1, dil=1, pad=1, seg2=(H2, H2, H2, H2))
    for mode in (0, 1):
        ref = torch.empty(M, C, device=dev, dtype=tdt)
        L.conv_igemm(x, w, ref, mode=mode, **geo)
        out, bad = torch.empty_like(ref), 0
        for _ in range(reps):
            out.fill_(float("nan")); L.conv_igemm(x, w, out, mode=mode, **geo)
            bad += int(not torch.equal(out, ref))
        print(f"3x3 512->512 mode {mode}: {bad} / {reps} launches differ from the first")


if __name__ == "__main__":
    main()
