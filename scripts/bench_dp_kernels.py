#!/usr/bin/env python
"""The data-parallel critical path at world-8 sizes, timed on ONE GPU with synthetic gathered buffers:
the kernels that sit on the main stream between an all-gather and the next launch of a data-parallel step —
  intra_weights_global  (hard-pixel order statistics over the gathered records, contrast_train.py:302-334)
  proto_merge           (global top-32 per class from the gathered candidates, :202-203)
for world = 1, 2, 4, 8 at the step's real P = 4096 contrast pixels per rank and view, plus the two all-gather buffer layouts the step
uses (one view's blocks a rank stride apart inside the [world][2 views][...] buffer).  Prints a table; profiles/r03_dp_kernels.txt."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wseg_amd import _lib as L                      # noqa: E402
from wseg_amd.loss_hip import _CAND_K, _CAND_L      # noqa: E402


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3          # us per call


def main():
    dev = "cuda"
    P, K = 4096, _CAND_K
    print(f"P = {P} contrast pixels per rank and view, K = {K}; us per launch (mean of 50 back-to-back launches)")
    print(f"{'world':>5} {'records':>8} {'intra_weights_global':>22} {'  (view 2 layout)':>18} {'proto_merge':>12} {'  (strided)':>12}")
    for world in (1, 2, 4, 8):
        g = torch.Generator().manual_seed(world)
        # realistic label mix: background ~45 % of the pixels, the rest over 3 present classes, similarities in (0.2, 1)
        y = torch.where(torch.rand(world, P, generator=g) < 0.45, torch.zeros(world, P), torch.randint(1, 21, (world, P), generator=g).float() % 3 * 5 + 2)
        rec = torch.empty(world, 2, 3, P)
        for v in range(2):
            rec[:, v, 0] = y.int().view(torch.float32)
            rec[:, v, 1] = torch.rand(world, P, generator=g) * 0.8 + 0.2
            rec[:, v, 2] = torch.rand(world, P, generator=g)
        rec = rec.to(dev)
        w = torch.empty(P, device=dev)
        t_i0 = timeit(lambda: L.intra_weights_global(rec.view(-1), w, P, world, world - 1, float(world), 6 * P))
        t_i1 = timeit(lambda: L.intra_weights_global(rec.view(-1)[3 * P:], w, P, world, 0, float(world), 6 * P))
        gathered = torch.zeros(world, 2, _CAND_L, device=dev)
        gathered[:, :, :21 * K] = torch.rand(world, 2, 21 * K, generator=g).to(dev)
        gathered[:, :, 21 * K:21 * K * 129] = torch.randn(world, 2, 21 * K * 128, generator=g).to(dev)
        protos = torch.empty(21, 128, device=dev)
        base = gathered.view(-1)[_CAND_L:]
        t_m1 = timeit(lambda: L.proto_merge(base, base[21 * K:], base[21 * K * 129:].view(torch.int32), protos, world, K, 2 * _CAND_L))
        cv, cf = gathered[:, 0, :21 * K].contiguous(), gathered[:, 0, 21 * K:21 * K * 129].contiguous()
        cc = torch.zeros(world, 21, device=dev, dtype=torch.int32)
        t_m0 = timeit(lambda: L.proto_merge(cv, cf, cc, protos, world, K))
        print(f"{world:>5} {world * P:>8} {t_i0:>22.1f} {t_i1:>18.1f} {t_m0:>12.1f} {t_m1:>12.1f}")


if __name__ == "__main__":
    main()
