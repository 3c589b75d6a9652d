"""CPU restatement of the multi-scale CAM inference (test infrastructure, see oracle/__init__.py).

contrast_infer.py:58-99: 8 forwards (4 scales x {orig, h-flip}), output #2 of the Net (the
PCM-refined CAM), bilinear resize to the original size with align_corners=False, label gating,
un-flip, sum, clamp, per-class min/max normalisation, argmax against a constant bg score.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import net as onet


def infer_one(img_list, label20, sd, orig_size, alpha=0.26):
    """img_list: 8 tensors [1,3,h_s,w_s] in VOC12ClsDatasetMSF order (voc12/data.py:100-121):
    [s0, flip(s0), s1, flip(s1), ...].  Returns (norm_cam [20,H,W] float32, pred [H,W] uint8,
    cam_dict {class->map})."""
    cams = []
    lab = label20.clone().view(20, 1, 1).numpy()
    for i, img in enumerate(img_list):
        with torch.no_grad():
            _, cam, _, _ = onet.net_forward(img, sd, None)
            cam = F.interpolate(cam[:, 1:, :, :], orig_size, mode="bilinear", align_corners=False)[0]
            cam = cam.numpy() * lab
            if i % 2 == 1:
                cam = np.flip(cam, axis=-1)
            cams.append(cam)
    return postprocess(cams, label20, alpha)


def postprocess(cam_list, label20, alpha=0.26):
    """contrast_infer.py:75-98."""
    sum_cam = np.sum(cam_list, axis=0)
    sum_cam[sum_cam < 0] = 0
    cam_max = np.max(sum_cam, (1, 2), keepdims=True)
    cam_min = np.min(sum_cam, (1, 2), keepdims=True)
    sum_cam[sum_cam < cam_min + 1e-5] = 0
    norm_cam = (sum_cam - cam_min - 1e-5) / (cam_max - cam_min + 1e-5)
    cam_dict = {}
    for i in range(20):
        if label20[i] > 1e-5:
            cam_dict[i] = norm_cam[i]
    bg_score = [np.ones_like(norm_cam[0]) * alpha]
    pred = np.argmax(np.concatenate((bg_score, norm_cam)), 0).astype(np.uint8)
    return norm_cam, pred, cam_dict
