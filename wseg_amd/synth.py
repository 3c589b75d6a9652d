"""Deterministic synthetic weights, images and labels for the hot path.

The reference needs the ImageNet ResNet-38 `.params` file (contrast_train.py:98-106) and the
VOC images; neither exists offline, so every test / benchmark regenerates *procedural*
values.  Every tensor is a closed-form integer hash of (name, seed, element index), computed
with exact int64 tensor arithmetic — bit-identical on CPU and on the GPU, any torch version.
"""
import zlib
from collections import OrderedDict

import torch

from .arch import state_dict_spec


def hash_uniform(key, n, device="cpu"):
    """float32 [n], uniform in [0,1) with 24 random bits: two rounds of a 32-bit
    xorshift-multiply mix of (index + key)."""
    h = torch.arange(n, dtype=torch.int64, device=device)
    h += int(key) & 0xFFFFFFFF
    h &= 0xFFFFFFFF
    for _ in range(2):
        h ^= (h >> 16)
        h *= 0x45D9F3B
        h &= 0xFFFFFFFF
    h ^= (h >> 16)
    h >>= 8
    return h.to(torch.float32).mul_(2.0 ** -24)


def _key(name, seed):
    return (zlib.crc32(name.encode()) + 0x9E3779B1 * (seed + 1)) & 0xFFFFFFFF


def _uniform(name, seed, shape, lo, hi, device="cpu"):
    n = 1
    for s in shape:
        n *= s
    u = hash_uniform(_key(name, seed), n, device)
    return u.mul_(float(hi - lo)).add_(float(lo)).reshape(shape)


def procedural_tensor(name, shape, seed=0, device="cpu"):
    """Value of state_dict entry `name` (float32, or int64 for num_batches_tracked)."""
    if name.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.int64, device=device)
    if name.endswith("running_var"):
        return _uniform(name, seed, shape, 0.5, 1.5, device)
    if name.endswith("running_mean"):
        return _uniform(name, seed, shape, -0.2, 0.2, device)
    if ".bn" in name or name.startswith("bn7"):
        if name.endswith(".weight"):
            return _uniform(name, seed, shape, 0.5, 1.5, device)
        return _uniform(name, seed, shape, -0.2, 0.2, device)
    # conv weights: kaiming-like; the closing conv of every residual branch is damped so the
    # activation scale stays O(1..30) through the 17 pre-activation blocks.
    cout, cin, kh, kw = shape
    fan_in = cin * kh * kw
    gain = 1.0
    if name.endswith("conv_branch2b1.weight") and not name.startswith(("b6.", "b7.")):
        gain = 0.5
    if name.endswith("conv_branch2b2.weight"):
        gain = 0.5
    if name.startswith("f9."):
        gain = 2.0
    if name.startswith("fc8."):
        gain = 0.03                           # keeps the GAP logits O(1): BCE not saturated
    a = gain * (6.0 / fan_in) ** 0.5          # uniform(-a,a) has std gain*sqrt(2/fan_in)
    return _uniform(name, seed, shape, -a, a, device)


def procedural_state_dict(seed=0, device="cpu"):
    sd = OrderedDict()
    for k, shape in state_dict_spec().items():
        sd[k] = procedural_tensor(k, shape, seed, device)
    return sd


def synthetic_images(n, size, seed=0, device="cpu"):
    """float32 [n,3,H,W], zero mean / unit variance, bell-shaped (sum of 3 uniforms) — the
    statistics of a normalised VOC crop (contrast_train.py:64-75, resnet38d.py:104-118)."""
    h, w = (size, size) if isinstance(size, int) else size
    u = sum(_uniform(f"img{j}", seed, (n, 3, h, w), -1.0, 1.0, device) for j in range(3))
    return u


def synthetic_labels(n, seed=0, device="cpu"):
    """float32 [n,20] multi-hot with 1-3 positives (VOC histogram: SURVEY.md §8d)."""
    u = _uniform("labels", seed, (n, 4), 0.0, 1.0).tolist()
    lab = torch.zeros(n, 20)
    for i in range(n):
        k = 1 if u[i][0] < 0.62 else (2 if u[i][0] < 0.91 else 3)
        first = int(u[i][1] * 20)
        for j in range(k):
            lab[i, (first + j * (1 + int(u[i][2 + (j > 1)] * 6))) % 20] = 1.0
    return lab.to(device)


DROPOUT_SITES = OrderedDict([("b6.dropout_2b1", (512, 0.3)), ("b6.dropout_2b2", (1024, 0.3)),
                             ("b7.dropout_2b1", (1024, 0.5)), ("b7.dropout_2b2", (2048, 0.5)),
                             ("dropout7", (4096, 0.5))])


def synthetic_dropout_masks(n, seed=0, device="cpu"):
    """Per-(n,channel) Dropout2d scale factors (0 or 1/(1-p)) for the five Dropout2d sites
    (resnet38d.py:64,68 with p=0.3 / 0.5; resnet38_contrast.py:14 with p=0.5)."""
    out = OrderedDict()
    for k, (c, p) in DROPOUT_SITES.items():
        keep = (_uniform("mask." + k, seed, (n, c), 0.0, 1.0, device) >= p).float() / (1.0 - p)
        out[k] = keep
    return out
