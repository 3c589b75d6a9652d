"""Pixel-to-prototype similarity + InfoNCE kernels (csrc/loss.hip: nce_sims, nce_loss_grad): time and achieved
HBM bandwidth at the reference shape (P = 4096) and a scaled sweep that shows the asymptotic rate.
Algorithmic bytes:  nce_sims      reads P*128*4 (features), writes P*128*4 (fn) + P*4 (norm) + 2*P*21*4 (similarities)
                    nce_loss_grad reads P*128*4 (fn) + 2*P*21*4 + P*(4+4+4+4), writes P*128*4 (dF)   (+ 2*21*128*4 prototypes each)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import _lib as L
dev = "cuda"
def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3     # us
print(f"{'P':>9s} {'sims us':>9s} {'GB/s':>8s} {'%8TB/s':>7s} | {'loss+grad us':>12s} {'GB/s':>8s} {'%8TB/s':>7s}")
for lg in (12, 14, 16, 18, 20, 22):
    P = 1 << lg
    F = torch.randn(P, 128, device=dev); pa = torch.nn.functional.normalize(torch.randn(21, 128, device=dev), dim=1); pb = torch.nn.functional.normalize(torch.randn(21, 128, device=dev), dim=1)
    fn = torch.empty_like(F); nrm = torch.empty(P, device=dev); So = torch.empty(P, 21, device=dev); St = torch.empty(P, 21, device=dev)
    y = torch.randint(0, 21, (P,), device=dev, dtype=torch.int32); y2 = torch.randint(0, 21, (P,), device=dev, dtype=torch.int32)
    w = torch.rand(P, device=dev) / P; dF = torch.empty_like(F); sums = torch.zeros(3, device=dev)
    it = 200 if lg <= 16 else 20
    t1 = timeit(lambda: L.nce_sims(F, pa, pb, fn, nrm, So, St, P), it)
    t2 = timeit(lambda: L.nce_loss_grad(fn, nrm, So, St, y, y2, w, pa, pb, dF, sums, P, 0.1 / (2 * P), 0.05), it)
    b1 = P * 128 * 4 * 2 + P * 4 + 2 * P * 21 * 4 + 2 * 21 * 128 * 4
    b2 = P * 128 * 4 * 2 + 2 * P * 21 * 4 + P * 16 + 2 * 21 * 128 * 4
    print(f"{P:9d} {t1:9.2f} {b1/t1/1e3:8.1f} {b1/t1/1e3/8000*100:6.1f}% | {t2:12.2f} {b2/t2/1e3:8.1f} {b2/t2/1e3/8000*100:6.1f}%", flush=True)
