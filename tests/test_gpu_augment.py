"""Device-side training augmentation (wseg_amd/augment.py + csrc/augment.hip) against the host pipeline of
wseg_amd/data.py (PIL) — the restatement of contrast_train.py:64-75 — from the same random draws: bit for bit."""
import random

import numpy as np
import PIL.Image
import pytest
import torch


def _image(h, w, seed):
    rng = np.random.default_rng(seed)
    low = rng.integers(0, 256, (h // 6 + 1, w // 6 + 1, 3), dtype=np.uint8)
    img = np.asarray(PIL.Image.fromarray(low).resize((w, h), PIL.Image.Resampling.BICUBIC)).copy()
    img[::7, ::5] = rng.integers(0, 256, img[::7, ::5].shape, dtype=np.uint8)        # some high-frequency content
    return img


def _host(img_u8, seed, crop):
    from wseg_amd import data as wdata
    from wseg_amd.resnet38_contrast import Normalize
    model_stub = type("M", (), {"normalize": Normalize()})()
    random.seed(seed)
    x = PIL.Image.fromarray(img_u8)
    for t in wdata.train_transform(model_stub, crop):
        x = t(x)
    return x, random.getstate()


SIZES = [(375, 500), (500, 375), (333, 500), (500, 500), (281, 500), (120, 160)]


def test_parameter_draws_and_coefficient_tables_follow_the_host_pipeline():
    """CPU: draw_params consumes Python's `random` exactly as the host transform chain does (same generator state afterwards),
    and the coefficient tables reproduce PIL's bicubic resize through a plain integer evaluation."""
    from wseg_amd import augment as A
    for si, (h, w) in enumerate(SIZES):
        img = _image(h, w, si)
        _, state = _host(img, 100 + si, 448)
        random.seed(100 + si)
        p = A.draw_params(w, h, 448)
        assert random.getstate() == state
        ref = np.asarray(PIL.Image.fromarray(img).resize((p["rw"], p["rh"]), resample=PIL.Image.Resampling.BICUBIC))
        (xb, xk), (yb, yk) = A.pil_bicubic_coeffs(w, p["rw"]), A.pil_bicubic_coeffs(h, p["rh"])

        def one_pass(src, b, k, axis):
            acc = np.full((src.shape[0], b.shape[0], 3) if axis == 1 else (b.shape[0], src.shape[1], 3), 1 << 21, np.int64)
            for j in range(k.shape[1]):
                idx = np.minimum(b[:, 0] + j, src.shape[axis] - 1)
                kj = np.where(j < b[:, 1], k[:, j], 0).astype(np.int64)
                acc += (src[:, idx, :].astype(np.int64) * kj[None, :, None]) if axis == 1 else (src[idx, :, :].astype(np.int64) * kj[:, None, None])
            return np.clip(acc >> 22, 0, 255).astype(np.uint8)
        got = one_pass(one_pass(img, xb, xk, 1), yb, yk, 0)
        assert np.array_equal(got, ref), (h, w, p["rw"], p["rh"])


@pytest.mark.gpu
@pytest.mark.parametrize("crop", [448, 128])
def test_device_augmentation_equals_the_host_pipeline(crop):
    from wseg_amd import augment as A
    aug = A.DeviceAugment("cuda", crop)
    samples, refs = [], []
    for si, (h, w) in enumerate(SIZES * 2):
        img = _image(h, w, 10 + si)
        seed = 500 + si
        ref, _ = _host(img, seed, crop)
        random.seed(seed)
        samples.append(A.make_sample("n%d" % si, img, np.eye(20, dtype=np.float32)[si % 20], crop))
        refs.append(ref)
    out, lab = aug(A.collate(samples))
    torch.cuda.synchronize()
    assert tuple(out.shape) == (len(samples), 3, crop, crop) and tuple(lab.shape) == (len(samples), 20)
    ops_seen = set()
    for i, (s, ref) in enumerate(zip(samples, refs)):
        got = out[i].cpu()
        ops_seen.add(tuple(s["params"]["op"]))
        assert torch.equal(got, ref), (i, s["params"], float((got - ref).abs().max()), int((got != ref).sum()))
    assert len(ops_seen) > 3                                   # several jitter orders were exercised
    assert any(s["params"]["cont_top"] > 0 or s["params"]["cont_left"] > 0 for s in samples) or crop == 128     # the zero-pad placement path
