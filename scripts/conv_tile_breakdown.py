#!/usr/bin/env python
"""Where a 256-tile conv workgroup spends its time (probe build: `WSEG_PROBES=1 bash wseg_amd/csrc/build.sh`): in-kernel s_memrealtime stamps
per workgroup — gather set-up, first-tile DMA wait, main loop, epilogue — for the layer shapes that dominate the step, both views batched, with
the epilogue forms the network uses (A: BN-ReLU second output only; B: raw output + BN-ReLU second output + residual; D: data gradient with
mask + residual).  Prints medians in microseconds and the dispersion of the tiles' end times per round.

    python scripts/conv_tile_breakdown.py [--only=512]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wseg_amd import _lib as L   # noqa: E402

SHAPES = [("512->512 3x3 56^2", 56, 512, 512, 3, 1), ("256->256 3x3 112^2", 112, 256, 256, 3, 1), ("1024->512 3x3 d2 56^2", 56, 1024, 512, 3, 2),
          ("1024->2048 3x3 d4 56^2", 56, 1024, 2048, 3, 4), ("2048->1024 1x1 56^2", 56, 2048, 1024, 1, 1)]


def main():
    dev, N = "cuda", 16
    only = [a[7:] for a in sys.argv if a.startswith("--only=")]
    for a in sys.argv:
        if a.startswith("--diag="):      # probe-only timing switches (wrong results): 1 = A requested every 9th K-tile, 2 = no B requests, 4 = no A requests
            L.debug_set_diag(int(a[7:]))
            print("diag", a[7:])
    for name, H, IC, OC, k, d in SHAPES:
        if only and not any(o in name for o in only):
            continue
        pad = d * (k // 2)
        H2 = H * 128 // 448
        M = N * (H * H + H2 * H2)
        x = torch.randn(M, IC, device=dev).bfloat16()
        w = (torch.randn(OC, k * k, IC, device=dev) * 0.02).bfloat16()
        if "--zero" in sys.argv:         # all-zero operands: the chip holds its top clock, so us per K-tile ranks loops by CYCLES (MI355X_MICROARCH.md, DVFS give-back)
            x.zero_(); w.zero_()
        if "--relu" in sys.argv:         # activations as the network has them: non-negative, half of them zero
            x.relu_()
        res = torch.randn(M, OC, device=dev).bfloat16()
        mask = torch.randn(M, OC, device=dev).bfloat16()
        out, out2 = torch.empty(M, OC, device=dev, dtype=torch.bfloat16), torch.empty(M, OC, device=dev, dtype=torch.bfloat16)
        sc, sh = torch.rand(OC, device=dev) + 0.5, torch.randn(OC, device=dev)
        geo = dict(N=N, IH=H, IW=H, IC=IC, OH=H, OW=H, OC=OC, KH=k, KW=k, stride=1, dil=d, pad=pad, seg2=(H2, H2, H2, H2))
        forms = {"A: out2 = relu(bn(.))": lambda: L.conv_igemm(x, w, None, out2, scale=sc, shift=sh, **geo),
                 "B: out + out2 + residual": lambda: L.conv_igemm(x, w, out, out2, r_post=res, scale=sc, shift=sh, **geo),
                 "D: dgrad, mask + residual": lambda: L.conv_igemm(x, w, out, None, mode=1, epi=1, scale=sc, mask=mask, r_post=res, **geo) if IC == OC else None}
        for fname, fn in forms.items():
            if fname.startswith("D") and IC != OC:
                continue
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            ntn = OC // 256
            t8, t7 = ((M + 255) // 256) * ntn, ((M + 223) // 224) * ntn
            nwg = t7 if ((t7 + 255) // 256) * 7 < ((t8 + 255) // 256) * 8 else t8
            raw = L.debug_stamps(nwg).astype(np.float64)
            st = raw[:, :7] / 100.0          # us
            t0 = st[:, 0].min()
            st -= t0
            setup, wait, loop, epi = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 4] - st[:, 3]
            epi_late = st[:, 6] - st[:, 5]
            order = np.argsort(st[:, 0])
            first = order[:min(256, nwg)]
            rest = order[min(256, nwg):]
            med = lambda a_: float(np.median(a_))
            nk = (k * k * IC) // 64
            flop = 2.0 * M * OC * IC * k * k
            print(f"{name:24s} {fname:28s} {ms * 1e3:7.1f} us launch ({flop / ms / 1e9:6.0f} TF/s), {nwg} tiles x {nk} K-tiles | per tile: set-up {med(setup):5.2f}  "
                  f"first-DMA wait {med(wait):5.2f}  main loop {med(loop):6.2f} ({med(loop) / nk:5.3f}/K-tile)  epilogue {med(epi):5.2f} (late waves {med(epi_late):5.2f}) us | "
                  f"round-1 tiles end at {med(st[first, 4]):6.1f} +- {float(np.std(st[first, 4])):4.1f} us"
                  + (f", later tiles start {med(st[rest, 0]):6.1f}, end {med(st[rest, 4]):6.1f} +- {float(np.std(st[rest, 4])):4.1f} us; last end {float(st[:, 4].max()):6.1f}" if len(rest) else ""), flush=True)
            if raw[:, 8:10].all() and not raw[:, 10:].any():   # WSEG_PROBES=1 build: shader-clock stamps around the main loop of wave 0
                cyc = raw[:, 9] - raw[:, 8]
                print(f"    main loop: {med(cyc) / nk:6.0f} cycles per K-tile at {med(cyc / (loop * 1e-6)) / 1e9:5.3f} GHz", flush=True)
            elif raw[:, 8:].any():             # WSEG_PROBES=2 build: cycles per K-tile in each slot of the main loop, waves 0 (early group) and 4 (late group)
                names = ["read1", "bar", "mfma1", "bar", "read2+wait", "bar", "mfma2", "bar"] if os.environ.get("WSEG_CONV_LOOP", "1") == "0" else \
                        ["compute", "dma-wait", "barrier", "-", "-", "-", "-", "-"]
                for wv, lo in ((0, 8), (4, 16)):
                    cyc = np.median(raw[:, lo:lo + 8], axis=0) / nk
                    print(f"    wave {wv}: cycles per K-tile  " + "  ".join(f"{n_} {c:6.0f}" for n_, c in zip(names, cyc)) + f"   total {cyc.sum():6.0f}", flush=True)


if __name__ == "__main__":
    main()
