"""Pixel-to-prototype similarity + InfoNCE (contrast_train.py:245-334): time and achieved HBM bandwidth at the reference shape
(P = 4096 pixels per view, D = 128, C = 21) and a scaled sweep that shows the asymptotic rate.

Bandwidth is ALGORITHMIC bytes / time, per launch, both views in one launch (2P pixels):
  nce_records  (forward half: similarities -> hard-pixel records)   P*(128*4 features + 4 label + 4 key)  read, P*12 written  = 532 B / pixel
  nce_fused    (similarities + 3 InfoNCE terms + gradient)          P*(128*4 + 4 + 4 + 4) read, P*128*4 dF written           = 1036 B / pixel
  (+ 2 x 21 x 128 x 4 B of prototypes per view, negligible)
`records x3` is the record pass with split-bf16 products (the bf16 / bf16x3 precision modes); the plain columns use the exact-f32 MFMA.  The unfused reference formulation (nce_sims + nce_loss_grad: normalised features and [P,21] similarity rows written to HBM and re-read)
is timed beside it and charged the SAME algorithmic bytes (1036 B / pixel for the pair), so the two columns compare like for like."""
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
print(f"{'P/view':>9s} | {'records us':>10s} {'GB/s':>8s} {'%8TB/s':>7s} | {'fused us':>9s} {'GB/s':>8s} {'%8TB/s':>7s} | {'records x3 us':>13s} {'GB/s':>8s} {'%8TB/s':>7s} | {'unfused sims+grad us':>20s} {'GB/s':>8s} {'%8TB/s':>7s}")
for lg in (12, 14, 16, 18, 20, 22):
    P = 1 << lg
    V = []
    for _ in range(2):
        F = torch.randn(P, 128, device=dev)
        V.append(dict(F=F, p=torch.nn.functional.normalize(torch.randn(21, 128, device=dev), dim=1), y=torch.randint(0, 21, (P,), device=dev, dtype=torch.int32),
                      w=torch.rand(P, device=dev) / P, dF=torch.empty_like(F), rkey=torch.rand(P, device=dev), rec=torch.empty(3, P, device=dev),
                      fn=torch.empty_like(F), nrm=torch.empty(P, device=dev), So=torch.empty(P, 21, device=dev), St=torch.empty(P, 21, device=dev)))
    sums = torch.zeros(3, device=dev)
    rec_views = [dict(F=v["F"], p_own=v["p"], y_own=v["y"], rkey=v["rkey"], rec=v["rec"]) for v in V]
    fus_views = [dict(F=v["F"], p_own=v["p"], p_oth=o["p"], y_own=v["y"], y_oth=o["y"], w_intra=v["w"], dF=v["dF"]) for v, o in ((V[0], V[1]), (V[1], V[0]))]
    it = 200 if lg <= 16 else 20
    t0 = timeit(lambda: L.nce_records(rec_views, P), it)
    t1 = timeit(lambda: L.nce_fused(fus_views, P, 0.1 / (2 * P), 0.05, sums), it)
    def unfused():
        for v, o in ((V[0], V[1]), (V[1], V[0])):
            L.nce_sims(v["F"], v["p"], o["p"], v["fn"], v["nrm"], v["So"], v["St"], P)
            L.nce_loss_grad(v["fn"], v["nrm"], v["So"], v["St"], v["y"], o["y"], v["w"], v["p"], o["p"], v["dF"], sums, P, 0.1 / (2 * P), 0.05)
    t2 = timeit(unfused, it)
    t3 = timeit(lambda: L.nce_records(rec_views, P, split_bf16=True), it)
    b0 = 2 * (P * 532 + 21 * 128 * 4)
    b1 = 2 * (P * 1036 + 2 * 21 * 128 * 4)
    print(f"{P:9d} | {t0:10.2f} {b0/t0/1e3:8.1f} {b0/t0/1e3/8000*100:6.1f}% | {t1:9.2f} {b1/t1/1e3:8.1f} {b1/t1/1e3/8000*100:6.1f}% | {t3:13.2f} {b0/t3/1e3:8.1f} {b0/t3/1e3/8000*100:6.1f}% | "
          f"{t2:20.2f} {b1/t2/1e3:8.1f} {b1/t2/1e3/8000*100:6.1f}%", flush=True)
