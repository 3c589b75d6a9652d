"""Device-side training augmentation: the transform chain of contrast_train.py:64-75 — RandomResizeLong(448, 768),
RandomHorizontalFlip, ColorJitter(0.3, 0.3, 0.3, 0.1), Normalize, RandomCrop(448), HWC_to_CHW — with the arithmetic on the GPU
(csrc/augment.hip) and only the JPEG decode and the random draws left on the host.

The host pipeline (wseg_amd/data.py, PIL) costs 45-60 ms of a core per image; a DataLoader worker of THIS pipeline decodes the file,
draws the parameters from Python's `random` in exactly the order the host transforms draw them, and builds the two small coefficient
tables of Pillow's bicubic resampler (float64, as Resample.c computes them).  `DeviceAugment` then turns a list of such samples into
the `[N, 3, crop, crop]` float32 batch on the device, bit for bit what the host pipeline produces from the same draws
(tests/test_gpu_augment.py).  The reference's own torchvision transforms are not importable offline: parity with THEM is unpinned.
"""
import ctypes as C
import math
import random

import numpy as np
import PIL.Image
import torch
from torch.utils.data import Dataset

from . import _lib as L
from . import data as wdata

PRECISION_BITS = 32 - 8 - 2          # Pillow Resample.c


def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def pil_bicubic_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (support 2, antialiased when shrinking), in float64
    with Resample.c's operation order: returns (bounds int32 [out, 2] = (first source index, count), coeffs int32 [out, ksize])."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size) + 0.5) * scale
    ss = 1.0 / fscale
    lo = center - support + 0.5
    xmin = np.where(lo < 0, 0, lo.astype(np.int64))
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize)[None, :]
    w = _bicubic((x + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(x < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size)
    for j in range(ksize):                                  # (sequential sum, as the C loop)
        ww = ww + w[:, j]
    w = np.where(ww[:, None] != 0.0, w / ww[:, None], w)
    kk = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
    return np.stack([xmin, xmax], axis=1).astype(np.int32), kk.astype(np.int32)


def draw_params(w, h, crop=448, min_long=448, max_long=768, jitter=(0.3, 0.3, 0.3, 0.1), rng=random):
    """The random draws of the host transform chain (wseg_amd/data.py: RandomResizeLong, RandomHorizontalFlip, ColorJitter,
    RandomCrop), in its order, from the same generator calls."""
    target_long = rng.randint(min_long, max_long)
    rw, rh = (int(round(w * target_long / h)), target_long) if w < h else (target_long, int(round(h * target_long / w)))
    flip = rng.random() < 0.5
    b, c, s, hj = jitter
    fb, fc, fs = rng.uniform(1 - b, 1 + b), rng.uniform(1 - c, 1 + c), rng.uniform(1 - s, 1 + s)
    hf = rng.uniform(-hj, hj)
    order = [0, 1, 2, 3]
    rng.shuffle(order)                                      # (the host shuffles its list of four ops: same permutation)
    w_space, h_space = rw - crop, rh - crop
    if w_space > 0:
        cont_left, img_left = 0, rng.randrange(w_space + 1)
    else:
        cont_left, img_left = rng.randrange(-w_space + 1), 0
    if h_space > 0:
        cont_top, img_top = 0, rng.randrange(h_space + 1)
    else:
        cont_top, img_top = rng.randrange(-h_space + 1), 0
    return dict(rw=rw, rh=rh, flip=int(flip), op=order, factor=[(fb, fc, fs, 0.0)[o] for o in order], hue_shift=int(hf * 255),
                cont_top=cont_top, cont_left=cont_left, img_top=img_top, img_left=img_left, ch=min(crop, rh), cw=min(crop, rw))


class VOC12ClsDatasetRaw(Dataset):
    """voc12/data.py:58-90 for the device pipeline: (name, decoded uint8 HWC image, label, params + coefficient tables)."""

    def __init__(self, img_name_list_path, voc12_root, labels_path, crop=448):
        self.img_name_list = wdata.load_img_name_list(img_name_list_path)
        self.voc12_root, self.crop = voc12_root, crop
        self.label_list = wdata.load_labels(labels_path, self.img_name_list)

    def __len__(self):
        return len(self.img_name_list)

    def __getitem__(self, idx):
        name = self.img_name_list[idx]
        img = np.asarray(PIL.Image.open(wdata.get_img_path(name, self.voc12_root)).convert("RGB"))
        return make_sample(name, img, self.label_list[idx], self.crop)                  # (packed per batch by `collate`)


def make_sample(name, img_u8, label, crop=448, rng=random):
    """One sample as a DataLoader worker produces it: numpy arrays only (they are packed per batch by `collate`)."""
    h, w = img_u8.shape[:2]
    p = draw_params(w, h, crop, rng=rng)
    xb, xk = pil_bicubic_coeffs(w, p["rw"])
    yb, yk = pil_bicubic_coeffs(h, p["rh"])
    p.update(H=h, W=w, xks=xk.shape[1], yks=yk.shape[1])
    return dict(name=name, img=img_u8, label=np.asarray(label, np.float32), params=p, xb=xb, xk=xk, yb=yb, yk=yk)


def collate(samples):
    """Runs in the DataLoader worker.  Images differ in size, so a batch is ONE uint8 blob (the decoded images back to back, 16-byte
    aligned), ONE int32 blob (the four coefficient tables of every image) and the per-image parameter dicts with their offsets — three
    tensors cross the process boundary instead of six per image."""
    n_img = n_tab = 0
    for s in samples:
        p = s["params"]
        p["img_off"] = n_img
        n_img += (s["img"].size + 15) // 16 * 16
        p["tab_off"] = []
        for k in ("xb", "xk", "yb", "yk"):
            p["tab_off"].append(n_tab)
            n_tab += s[k].size
    img = np.empty(n_img, np.uint8)
    tab = np.empty(n_tab, np.int32)
    for s in samples:
        p = s["params"]
        img[p["img_off"]:p["img_off"] + s["img"].size] = s["img"].reshape(-1)
        for k, o in zip(("xb", "xk", "yb", "yk"), p["tab_off"]):
            tab[o:o + s[k].size] = s[k].reshape(-1)
    return dict(img=torch.from_numpy(img), tab=torch.from_numpy(tab), label=torch.from_numpy(np.stack([s["label"] for s in samples])),
                params=[s["params"] for s in samples], names=[s["name"] for s in samples])


class AugDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("rh", C.c_int32), ("rw", C.c_int32),
                ("xb", C.c_void_p), ("xk", C.c_void_p), ("xks", C.c_int32), ("yb", C.c_void_p), ("yk", C.c_void_p), ("yks", C.c_int32),
                ("tmp", C.c_void_p), ("img", C.c_void_p), ("flip", C.c_int32), ("op", C.c_int32 * 4), ("factor", C.c_float * 4),
                ("hue_shift", C.c_int32), ("cont_top", C.c_int32), ("cont_left", C.c_int32), ("img_top", C.c_int32),
                ("img_left", C.c_int32), ("ch", C.c_int32), ("cw", C.c_int32), ("out", C.c_void_p)]


L.lib.wseg_sizeof_aug_desc.restype = C.c_size_t
if L.lib.wseg_sizeof_aug_desc() != C.sizeof(AugDesc):
    raise ImportError(f"wseg_aug_desc: library {L.lib.wseg_sizeof_aug_desc()} bytes, binding {C.sizeof(AugDesc)}: rebuild libwseg_hip.so")


def normalize_lut(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """float32((v / 255. - mean) / std) for v = 0..255 per channel — the values network/resnet38d.py:104-118 produces (float64 arithmetic,
    stored to float32)."""
    v = np.arange(256, dtype=np.float64)
    return np.stack([((v / 255. - mean[c]) / std[c]).astype(np.float32) for c in range(3)])


class DeviceAugment:
    def __init__(self, device, crop=448):
        self.device, self.crop = torch.device(device), crop
        self.lut = torch.from_numpy(normalize_lut()).to(self.device).contiguous()
        self._ring, self._slot = [], 0                       # page-locked descriptor staging: a pageable copy would make the host wait for the
                                                             # stream (= the whole previous training step) on every call

    def __call__(self, batch):
        """batch: what `collate` made of a list of samples (a list of samples is accepted too) -> (float32 [N, 3, crop, crop] on the
        device, labels [N, 20])."""
        if self.device.type != "cuda":
            raise RuntimeError("DeviceAugment runs on the MI355X only (the host pipeline is wseg_amd.data.train_transform)")
        if isinstance(batch, (list, tuple)):
            batch = collate(batch)
        params, crop, dev = batch["params"], self.crop, self.device
        n = len(params)
        # (a DataLoader with pin_memory=True hands over page-locked blobs: its pinning thread did the copy in the background)
        d_img = (batch["img"] if batch["img"].is_pinned() else batch["img"].pin_memory()).to(dev, non_blocking=True)
        d_tab = (batch["tab"] if batch["tab"].is_pinned() else batch["tab"].pin_memory()).to(dev, non_blocking=True)
        tmp_off, res_off, total_tmp, total_res = [], [], 0, 0
        for p in params:
            tmp_off.append(total_tmp); total_tmp += (p["H"] * p["rw"] * 3 + 15) // 16 * 16
            res_off.append(total_res); total_res += (p["rh"] * p["rw"] * 3 + 15) // 16 * 16
        d_tmp = torch.empty(total_tmp, device=dev, dtype=torch.uint8)
        d_res = torch.empty(total_res, device=dev, dtype=torch.uint8)
        out = torch.empty(n, 3, crop, crop, device=dev, dtype=torch.float32)
        descs = (AugDesc * n)()
        max_pixels = 1
        for i, (p, mo, ro) in enumerate(zip(params, tmp_off, res_off)):
            d, to = descs[i], p["tab_off"]
            d.src, d.H, d.W, d.rh, d.rw = d_img.data_ptr() + p["img_off"], p["H"], p["W"], p["rh"], p["rw"]
            d.xb, d.xk, d.xks = d_tab.data_ptr() + 4 * to[0], d_tab.data_ptr() + 4 * to[1], p["xks"]
            d.yb, d.yk, d.yks = d_tab.data_ptr() + 4 * to[2], d_tab.data_ptr() + 4 * to[3], p["yks"]
            d.tmp, d.img, d.flip, d.hue_shift = d_tmp.data_ptr() + mo, d_res.data_ptr() + ro, p["flip"], p["hue_shift"]
            for j in range(4):
                d.op[j], d.factor[j] = p["op"][j], p["factor"][j]
            d.cont_top, d.cont_left, d.img_top, d.img_left, d.ch, d.cw = (p[k] for k in ("cont_top", "cont_left", "img_top", "img_left", "ch", "cw"))
            d.out = out[i].data_ptr()
            max_pixels = max(max_pixels, p["H"] * p["rw"], p["rh"] * p["rw"])
        nbytes = C.sizeof(descs)
        if len(self._ring) < 4 or self._ring[self._slot][0].numel() < nbytes:
            entry = (torch.empty(max(nbytes, 64 * C.sizeof(AugDesc)), dtype=torch.uint8).pin_memory(), torch.cuda.Event())
            if len(self._ring) < 4:
                self._ring.append(entry); self._slot = len(self._ring) - 1
            else:
                self._ring[self._slot] = entry
        else:
            self._ring[self._slot][1].synchronize()           # the copy that last used this slot has been consumed
        stage, ev = self._ring[self._slot]
        C.memmove(stage.data_ptr(), C.addressof(descs), nbytes)
        d_desc = stage[:nbytes].to(dev, non_blocking=True)
        ev.record()
        self._slot = (self._slot + 1) % 4
        sums = torch.empty(n * 4, device=dev, dtype=torch.int64)
        L.check(L.lib.wseg_augment_batch(C.c_void_p(d_desc.data_ptr()), n, max_pixels, C.c_void_p(self.lut.data_ptr()), crop,
                                         C.c_void_p(sums.data_ptr()), C.c_void_p(L.stream_ptr())), "wseg_augment_batch")
        self._keep = (d_img, d_tab, d_tmp, d_res, d_desc, sums)           # (alive until the next call: the kernels are asynchronous)
        return out, batch["label"].to(dev, non_blocking=True)

    def batches(self, loader_iter, overlap=False):
        """Iterate `loader_iter`.  overlap=False (default): every batch is copied and augmented on the caller's stream, in front of its
        training step (1.5 ms of small kernels + a 9 MB copy: measured 404 vs 425 images/s for the host pipeline on the same box, whose
        256 fast cores keep up with one GPU).  overlap=True runs batch i + 1 on a side stream while the caller's stream runs step i —
        measured SLOWER (373 vs 406 images/s on its box): the thousands of small workgroups interleave with the one-workgroup-per-CU
        conv tiles and desynchronise their rounds; kept as a switch for hosts where it pays."""
        if not overlap:
            for batch in loader_iter:
                yield self(batch)
            return
        side = getattr(self, "_side", None)
        if side is None:
            side = self._side = torch.cuda.Stream(self.device)

        def launch(batch):
            side.wait_stream(torch.cuda.current_stream(self.device))       # (scratch freed on the caller's stream may be reused here)
            with torch.cuda.stream(side):
                img, lab = self(batch)
                keep = self._keep
                ev = side.record_event()
            return img, lab, ev, keep

        def ready(item):
            img, lab, ev, keep = item
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            img.record_stream(cur); lab.record_stream(cur)
            return img, lab

        nxt = None
        for batch in loader_iter:
            item = launch(batch)
            if nxt is not None:
                yield ready(nxt)
            nxt = item
        if nxt is not None:
            yield ready(nxt)
