"""Host-side VOC12 plumbing for the CLIs: file formats of voc12/data.py:40-121 and the transforms of
contrast_train.py:64-75 / tool/imutils.py:6-67, restated on PIL >= 10 + numpy only (the reference
needs torchvision and the removed PIL.Image.CUBIC).  Not part of the accelerated path: it feeds it.

List files: one line per image, `/JPEGImages/<name>.jpg [/SegmentationClassAug/<name>.png]`; the image
name is characters [-15:-4] of the first field (voc12/data.py:49-55).  Labels: a pickled dict
name -> float32[20] (`cls_labels.npy`, voc12/data.py:40-44) — read by a restricted unpickler that admits numeric numpy
containers only (wseg_amd/safe_npy.py: the file is untrusted input); a plain .npz (names, labels) is accepted too.
"""
import os
import random

import numpy as np
import PIL.Image
import PIL.ImageEnhance
import torch
from torch.utils.data import Dataset

from .safe_npy import load_pickled_npy

IMG_FOLDER_NAME = "JPEGImages"
BICUBIC = PIL.Image.Resampling.BICUBIC


def img_name_of(line):
    """Image name of one list line: `voc12/*.txt` lines are `/JPEGImages/<name>.jpg [/SegmentationClassAug/<name>.png]`
    (voc12/data.py:49-55 takes characters [-15:-4] of the first field); the devkit's `ImageSets/Segmentation/*.txt` lists that
    eval.py:112 reads hold the bare name."""
    first = line.strip().split(' ')[0]
    if '/' in first or first.lower().endswith(('.jpg', '.png')):
        return os.path.splitext(os.path.basename(first))[0]
    return first


def load_img_name_list(dataset_path):
    return [img_name_of(line) for line in open(dataset_path).read().splitlines() if line.strip()]


def get_img_path(img_name, voc12_root):
    return os.path.join(voc12_root, IMG_FOLDER_NAME, img_name + '.jpg')


def load_labels(path, names):
    if path.endswith(".npz"):
        z = np.load(path)
        table = dict(zip([str(n) for n in z["names"]], z["labels"].astype(np.float32)))
    else:
        table = load_pickled_npy(path)
    return [np.asarray(table[n], np.float32) for n in names]


class RandomResizeLong:                                    # tool/imutils.py:6-27
    def __init__(self, min_long, max_long):
        self.min_long, self.max_long = min_long, max_long

    def __call__(self, img):
        target_long = random.randint(self.min_long, self.max_long)
        w, h = img.size
        shape = (int(round(w * target_long / h)), target_long) if w < h else (target_long, int(round(h * target_long / w)))
        return img.resize(shape, resample=BICUBIC)


class RandomHorizontalFlip:
    def __call__(self, img):
        return img.transpose(PIL.Image.Transpose.FLIP_LEFT_RIGHT) if random.random() < 0.5 else img


class ColorJitter:
    """brightness/contrast/saturation factors U(1-a,1+a), hue shift U(-h,h), random order
    (the torchvision transform the reference configures at contrast_train.py:68-69)."""

    def __init__(self, brightness=0.3, contrast=0.3, saturation=0.3, hue=0.1):
        self.b, self.c, self.s, self.h = brightness, contrast, saturation, hue

    def __call__(self, img):
        ops = []
        ops.append(lambda im, f=random.uniform(1 - self.b, 1 + self.b): PIL.ImageEnhance.Brightness(im).enhance(f))
        ops.append(lambda im, f=random.uniform(1 - self.c, 1 + self.c): PIL.ImageEnhance.Contrast(im).enhance(f))
        ops.append(lambda im, f=random.uniform(1 - self.s, 1 + self.s): PIL.ImageEnhance.Color(im).enhance(f))
        hf = random.uniform(-self.h, self.h)

        def hue(im):
            hsv = np.array(im.convert("HSV"))
            hsv[..., 0] = (hsv[..., 0].astype(np.int16) + int(hf * 255)) % 256
            return PIL.Image.fromarray(hsv, "HSV").convert("RGB")
        ops.append(hue)
        random.shuffle(ops)
        for op in ops:
            img = op(img)
        return img


class RandomCrop:                                          # tool/imutils.py:30-67
    def __init__(self, cropsize):
        self.cropsize = cropsize

    def __call__(self, imgarr):
        h, w, c = imgarr.shape
        cs = self.cropsize
        ch, cw = min(cs, h), min(cs, w)
        w_space, h_space = w - cs, h - cs
        if w_space > 0:
            cont_left, img_left = 0, random.randrange(w_space + 1)
        else:
            cont_left, img_left = random.randrange(-w_space + 1), 0
        if h_space > 0:
            cont_top, img_top = 0, random.randrange(h_space + 1)
        else:
            cont_top, img_top = random.randrange(-h_space + 1), 0
        container = np.zeros((cs, cs, c), np.float32)
        container[cont_top:cont_top + ch, cont_left:cont_left + cw] = imgarr[img_top:img_top + ch, img_left:img_left + cw]
        return container


def HWC_to_CHW(img):
    return np.transpose(img, (2, 0, 1))


class VOC12ClsDataset(Dataset):                            # voc12/data.py:58-90
    def __init__(self, img_name_list_path, voc12_root, labels_path, transform=None):
        self.img_name_list = load_img_name_list(img_name_list_path)
        self.voc12_root = voc12_root
        self.transform = transform
        self.label_list = load_labels(labels_path, self.img_name_list)

    def __len__(self):
        return len(self.img_name_list)

    def __getitem__(self, idx):
        name = self.img_name_list[idx]
        img = PIL.Image.open(get_img_path(name, self.voc12_root)).convert("RGB")
        if self.transform:
            for t in self.transform:
                img = t(img)
        return name, img, torch.from_numpy(self.label_list[idx])


class VOC12ClsDatasetMSF(VOC12ClsDataset):                 # voc12/data.py:92-121
    def __init__(self, img_name_list_path, voc12_root, labels_path, scales, inter_transform=None, unit=1):
        super().__init__(img_name_list_path, voc12_root, labels_path, transform=None)
        self.scales, self.unit, self.inter_transform = scales, unit, inter_transform

    def __getitem__(self, idx):
        name, img, label = super().__getitem__(idx)
        rounded = (int(round(img.size[0] / self.unit) * self.unit), int(round(img.size[1] / self.unit) * self.unit))
        out = []
        for s in self.scales:
            s_img = img.resize((round(rounded[0] * s), round(rounded[1] * s)), resample=BICUBIC)
            for t in (self.inter_transform or []):
                s_img = t(s_img)
            out.append(s_img)
            out.append(np.flip(s_img, -1).copy())
        return name, out, label


def train_transform(model, crop_size):
    """contrast_train.py:64-75."""
    return [RandomResizeLong(448, 768), RandomHorizontalFlip(), ColorJitter(0.3, 0.3, 0.3, 0.1), np.asarray,
            model.normalize, RandomCrop(crop_size), HWC_to_CHW, torch.from_numpy]
