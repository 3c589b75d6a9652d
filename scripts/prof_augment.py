"""Where a DeviceAugment call spends its time: host staging (pinning, descriptors) vs the 11 launches on the GPU."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, PIL.Image, torch
from wseg_amd import augment as A
rng = np.random.default_rng(0)
def image(h, w):
    low = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)
    return np.asarray(PIL.Image.fromarray(low).resize((w, h), PIL.Image.Resampling.BICUBIC))
random.seed(0)
batches = [A.collate([A.make_sample("x", image(375, 500) if i % 3 else image(500, 375), np.zeros(20, np.float32)) for i in range(16)]) for _ in range(4)]
aug = A.DeviceAugment("cuda", 448)
for b in batches: aug(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    for b in batches: aug(b)
t_host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pinned = [dict(b, img=b["img"].pin_memory(), tab=b["tab"].pin_memory(), label=b["label"].pin_memory()) for b in batches]
torch.cuda.synchronize()
t0 = time.perf_counter()
e0.record()
for _ in range(5):
    for b in pinned: aug(b)
e1.record()
t_host_p = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
print(f"unpinned input: host enqueue {t_host * 1e3:.2f} ms per batch, wall {t_all * 1e3:.2f} ms; pinned input: host enqueue {t_host_p * 1e3:.2f} ms, GPU stream time {e0.elapsed_time(e1) / 20:.2f} ms per batch of 16")
