"""CLI of the reference's contrast_infer.py (same flags, :19-31) on the MI355X path.  Writes the same files:
<out_cam>/<name>.npy (pickled dict class -> float32[H,W] of the present classes, :82-90) and
<out_cam_pred>/<name>.png (uint8 argmax, :97-99).  --out_crf is accepted and rejected: dense CRF needs
pydensecrf (absent offline, CPU-only post-process, out of the hot path)."""
import argparse
import importlib
import os

import numpy as np
import PIL.Image
import torch

from . import data as wdata
from . import synth
from .infer import infer_image


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--weights", required=True, type=str)
    parser.add_argument("--network", default="wseg_amd.resnet38_contrast", type=str)
    parser.add_argument("--infer_list", default="voc12/train.txt", type=str)
    parser.add_argument("--num_workers", default=8, type=int)
    parser.add_argument("--voc12_root", default='VOC2012', type=str)
    parser.add_argument("--out_cam", default=None, type=str)
    parser.add_argument("--out_crf", default=None, type=str)
    parser.add_argument("--out_cam_pred", default=None, type=str)
    parser.add_argument("--out_cam_pred_alpha", default=0.26, type=float)
    parser.add_argument("--crf_iters", default=10, type=float)
    parser.add_argument("--labels", default="voc12/cls_labels.npy", type=str)
    parser.add_argument("--precision", default=None, choices=[None, "bf16", "fp32", "bf16x3"])
    args = parser.parse_args(argv)
    if args.out_crf is not None:
        raise SystemExit("--out_crf needs pydensecrf, which is not available offline (out of the accelerated path)")

    Net = getattr(importlib.import_module(args.network), 'Net')
    model = Net(precision=args.precision) if args.precision else Net()
    if args.weights == "procedural":
        model.load_state_dict(synth.procedural_state_dict(0))
    else:
        model.load_state_dict(torch.load(args.weights, map_location="cpu", weights_only=True))
    model.eval()
    model.cuda()

    ds = wdata.VOC12ClsDatasetMSF(args.infer_list, args.voc12_root, args.labels, scales=[0.5, 1.0, 1.5, 2.0],
                                  inter_transform=[np.asarray, model.normalize, wdata.HWC_to_CHW])
    loader = torch.utils.data.DataLoader(ds, shuffle=False, num_workers=args.num_workers, pin_memory=True)
    for it, (img_name, img_list, label) in enumerate(loader):
        img_name, label = img_name[0], label[0]
        orig = np.asarray(PIL.Image.open(wdata.get_img_path(img_name, args.voc12_root)))
        norm_cam, pred, cam_dict = infer_image(model, img_list, label, orig.shape[:2], args.out_cam_pred_alpha)
        if args.out_cam is not None:
            os.makedirs(args.out_cam, exist_ok=True)
            np.save(os.path.join(args.out_cam, img_name + '.npy'), {k: v.cpu().numpy() for k, v in cam_dict.items()})
        if args.out_cam_pred is not None:
            os.makedirs(args.out_cam_pred, exist_ok=True)
            PIL.Image.fromarray(pred.cpu().numpy()).save(os.path.join(args.out_cam_pred, img_name + '.png'))


if __name__ == '__main__':
    main()
