"""End-to-end rate of `wseg_amd.contrast_train` on JPEG files: is the host augmentation pipeline (PIL decode ->
RandomResizeLong(448,768) bicubic -> flip -> ColorJitter -> normalise -> RandomCrop(448), contrast_train.py:64-75 restated in
wseg_amd/data.py) the limit beside a 437 img/s training step?  Writes N synthetic 500x375 JPEGs (smooth random fields, so the
files compress like photographs), then measures (a) the DataLoader alone at several worker counts, (b) loader + H2D copy + Trainer.step.

    python scripts/bench_data_pipeline.py [n_images=512] [workers=4,8,16]
"""
import contextlib, io, os, random, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import PIL.Image
import torch
from wseg_amd import augment as waug, data as wdata, synth
from wseg_amd.optim import PolyOptimizer
from wseg_amd.resnet38_contrast import Net
from wseg_amd.train import Trainer

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 512
workers = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,8,16").split(",")]
tmp = tempfile.mkdtemp()
root = os.path.join(tmp, "VOC2012"); os.makedirs(os.path.join(root, "JPEGImages"))
rng = np.random.default_rng(0)
names = ["2008_%06d" % i for i in range(n_img)]
for i, n in enumerate(names):
    h, w = (375, 500) if i % 3 else (500, 375)
    low = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)
    img = PIL.Image.fromarray(low).resize((w, h), PIL.Image.Resampling.BICUBIC)
    img.save(os.path.join(root, "JPEGImages", n + ".jpg"), quality=90)
lst = os.path.join(tmp, "list.txt")
open(lst, "w").write("\n".join(f"/JPEGImages/{n}.jpg /SegmentationClassAug/{n}.png" for n in names) + "\n")
np.savez(os.path.join(tmp, "labels.npz"), names=np.array(names), labels=synth.synthetic_labels(n_img, 0).numpy())
print(f"{n_img} JPEGs of 500x375, {sum(os.path.getsize(os.path.join(root, 'JPEGImages', n + '.jpg')) for n in names) / n_img / 1024:.0f} KB each; host cores: {os.cpu_count()}", flush=True)

model = Net(precision="bf16")
ds = wdata.VOC12ClsDataset(lst, root, os.path.join(tmp, "labels.npz"), wdata.train_transform(model, 448))
B = 16


def loader(nw):
    return torch.utils.data.DataLoader(ds, batch_size=B, shuffle=True, num_workers=nw, pin_memory=True, drop_last=True,
                                       persistent_workers=False, worker_init_fn=lambda wid: (np.random.seed(wid), random.seed(wid)))


for nw in workers:
    it = iter(loader(nw))
    next(it)                                      # workers started, first batch out
    t0 = time.perf_counter(); n = 0
    for pack in it:
        n += pack[1].shape[0]
    dt = time.perf_counter() - t0
    print(f"host pipeline alone, {nw:2d} workers: {n / dt:7.1f} images/s", flush=True)

dsr = waug.VOC12ClsDatasetRaw(lst, root, os.path.join(tmp, "labels.npz"), 448)


def loader_raw(nw):
    return torch.utils.data.DataLoader(dsr, batch_size=B, shuffle=True, num_workers=nw, drop_last=True, collate_fn=waug.collate, pin_memory=True,
                                       worker_init_fn=lambda wid: (np.random.seed(wid), random.seed(wid)))


for nw in workers:
    it = iter(loader_raw(nw))
    next(it)
    t0 = time.perf_counter(); n = 0
    for batch in it:
        n += len(batch["params"])
    dt = time.perf_counter() - t0
    print(f"decode + parameter draws only (device-augment workers), {nw:2d} workers: {n / dt:7.1f} images/s", flush=True)

if torch.cuda.is_available():
    with contextlib.redirect_stdout(io.StringIO()):
        groups = model.get_parameter_groups()
    lr = 1e-5
    opt = PolyOptimizer([{'params': groups[0], 'lr': lr, 'weight_decay': 5e-4}, {'params': groups[1], 'lr': 2 * lr, 'weight_decay': 0},
                         {'params': groups[2], 'lr': 10 * lr, 'weight_decay': 5e-4}, {'params': groups[3], 'lr': 20 * lr, 'weight_decay': 0}],
                        lr=lr, weight_decay=5e-4, max_step=10000)
    model.load_state_dict(synth.procedural_state_dict(0, device="cuda")); model.cuda(); model.train()
    tr = Trainer(model, opt, 0.20, random.Random(0), False)

    def run_ahead(batches):
        """steps over an iterator of (img, lab) device batches, one batch ahead (Trainer.step next_img1); returns the images stepped"""
        n, cur = 0, next(batches, None)
        while cur is not None:
            nxt = next(batches, None)
            tr.step(cur[0], cur[1], next_img1=nxt[0] if nxt is not None else None)
            n += cur[0].shape[0]
            cur = nxt
        return n

    for nw in workers:
        it = iter(loader(nw))
        pack = next(it)
        tr.step(pack[1].cuda(non_blocking=True), pack[2].cuda(non_blocking=True))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = run_ahead((pack[1].cuda(non_blocking=True), pack[2].cuda(non_blocking=True)) for pack in it)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"JPEG files -> training step (B = {B}, bf16), {nw:2d} workers: {n / dt:7.1f} images/s", flush=True)
    aug = waug.DeviceAugment("cuda", 448)
    for nw, overlap in [(w_, o_) for w_ in workers for o_ in (False, True)]:
        it = aug.batches(iter(loader_raw(nw)), overlap=overlap)
        img, lab = next(it)
        tr.step(img, lab)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = run_ahead(it)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"JPEG files -> device augmentation ({'one batch ahead on a side stream' if overlap else 'same stream'}) -> training step (B = {B}, bf16), {nw:2d} workers: {n / dt:7.1f} images/s", flush=True)
    it = iter(loader_raw(workers[-1]))
    batches = [next(it) for _ in range(4)]
    aug(batches[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        for b_ in batches:
            aug(b_)
    torch.cuda.synchronize()
    print(f"device augmentation alone (host staging + 11 launches per batch of {B}): {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per batch", flush=True)
