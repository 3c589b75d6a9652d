import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import _lib as L
dev="cuda"
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for name, N, H, IC, OC, k, d in [("512->512 3x3", 16, 56, 512, 512, 3, 1), ("1024->2048 3x3 d4", 16, 56, 1024, 2048, 3, 4), ("2048->4096 1x1", 16, 56, 2048, 4096, 1, 1)]:
    pad = d * (k // 2)
    x = torch.randn(N, H, H, IC, device=dev).bfloat16(); wf = (torch.randn(OC, k*k, IC, device=dev)*0.02).bfloat16()
    y = torch.empty(N, H, H, OC, device=dev, dtype=torch.bfloat16)
    flop = 2.0*N*H*H*OC*IC*k*k
    for hint, lab in [(0, "normal"), (-1, "A from zero page"), (-2, "B from zero page")]:
        t = timeit(lambda: L.conv_igemm(x, wf, y, N=N, IH=H, IW=H, IC=IC, OH=H, OW=H, OC=OC, KH=k, KW=k, dil=d, pad=pad, bm_hint=hint))
        print(f"{name:20s} {lab:18s} {t:7.3f} ms {flop/t/1e9:7.1f} TF/s", flush=True)
