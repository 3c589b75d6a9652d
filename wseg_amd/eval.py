"""CAM mIoU evaluation — counterpart of the reference's eval.py:13-86,109-136 (the second half of the headline
metric, "CAM mIoU vs ref").  Host-side numpy, like the reference (which forks 8 CPU processes); the per-image
TP/P/T counting is vectorised with bincount instead.  Inputs are the files contrast_infer writes:
  type 'png'  : <predict_folder>/<name>.png (uint8 argmax)
  type 'npy'  : <predict_folder>/<name>.npy, a pickled dict class->float32[H,W]; prediction = argmax over
                [bg threshold t] ++ maps (eval.py:31-39), swept over t in --curve mode (eval.py:130-136).
Ground truth: <gt_folder>/<name>.png, label 255 ignored (eval.py:41-43).  IoU_c = TP_c / (T_c + P_c - TP_c).

    python -m wseg_amd.eval --list voc12/val.txt --predict_dir out_cam --gt_dir VOC2012/SegmentationClassAug --type npy --t 0.26
"""
import argparse
import os

import numpy as np
import PIL.Image

from .data import load_img_name_list
from .safe_npy import load_pickled_npy

CATEGORIES = ['background', 'aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
              'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']


def load_prediction(folder, name, input_type, threshold, num_cls=21):
    if input_type == 'png':
        return np.array(PIL.Image.open(os.path.join(folder, name + '.png')))
    d = load_pickled_npy(os.path.join(folder, name + '.npy'))       # dict class -> float32[H,W] (restricted unpickler)
    h, w = list(d.values())[0].shape
    tensor = np.zeros((num_cls, h, w), np.float32)
    for key, v in d.items():
        tensor[key + 1] = v
    tensor[0, :, :] = threshold
    return np.argmax(tensor, axis=0).astype(np.uint8)


def do_eval(name_list, predict_folder, gt_folder, input_type='png', threshold=1.0, num_cls=21):
    TP = np.zeros(num_cls, np.int64); P = np.zeros(num_cls, np.int64); T = np.zeros(num_cls, np.int64)
    for name in name_list:
        predict = load_prediction(predict_folder, name, input_type, threshold, num_cls)
        gt = np.array(PIL.Image.open(os.path.join(gt_folder, name + '.png')))
        cal = gt < 255
        mask = (predict == gt) & cal
        P += np.bincount(predict[cal].astype(np.int64), minlength=num_cls)[:num_cls]
        T += np.bincount(gt[cal].astype(np.int64), minlength=num_cls)[:num_cls]
        TP += np.bincount(gt[mask].astype(np.int64), minlength=num_cls)[:num_cls]
    IoU = TP / (T + P - TP + 1e-10)
    T_TP = T / (TP + 1e-10)
    P_TP = P / (TP + 1e-10)
    FP_ALL = (P - TP) / (T + P - TP + 1e-10)
    FN_ALL = (T - TP) / (T + P - TP + 1e-10)
    out = {CATEGORIES[i]: IoU[i] * 100 for i in range(num_cls)}
    out['mIoU'] = float(np.mean(IoU) * 100)
    out['t_tp'] = float(np.mean(T_TP[1:])); out['p_tp'] = float(np.mean(P_TP[1:]))
    out['fp_all'] = float(np.mean(FP_ALL[1:])); out['fn_all'] = float(np.mean(FN_ALL[1:]))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", default='voc12/train.txt', type=str)
    ap.add_argument("--predict_dir", default="./out_rw", type=str)
    ap.add_argument("--gt_dir", default='VOC2012/SegmentationClassAug', type=str)
    ap.add_argument("--type", default='png', choices=['npy', 'png'], type=str)
    ap.add_argument("--t", default=None, type=float)
    ap.add_argument("--curve", action="store_true")
    a = ap.parse_args(argv)
    names = load_img_name_list(a.list)              # both list formats: voc12/*.txt path lines and the devkit's bare names
    if not a.curve:
        res = do_eval(names, a.predict_dir, a.gt_dir, a.type, a.t if a.t is not None else 1.0)
        print('mIoU: %.3f' % res['mIoU'])
        return res
    best = None
    for i in range(60):                                          # eval.py:130-136: t = 0.00 .. 0.59
        t = i / 100.0
        res = do_eval(names, a.predict_dir, a.gt_dir, a.type, t)
        print('%d/60 background score: %.3f\tmIoU: %.3f%%' % (i, t, res['mIoU']))
        if best is None or res['mIoU'] > best[1]:
            best = (t, res['mIoU'])
    print('best background score: %.3f\tmIoU: %.3f%%' % best)
    return best


if __name__ == '__main__':
    main()
