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
    # One image behind: the device work of image i is enqueued (nothing in infer_image synchronises), its outputs start their way to pinned host
    # buffers, and only then are the files of image i - 1 written — the png / npy writes and the loader hand-over overlap the GPU instead of
    # alternating with it (the reference's loop, contrast_infer.py:52-99, syncs per image on `.cpu()`).
    def start(img_name, pred, cam_dict):
        host_pred = torch.empty(pred.shape, dtype=pred.dtype, pin_memory=True)
        host_pred.copy_(pred, non_blocking=True)
        host_cams = None
        if args.out_cam is not None:
            host_cams = {}
            for k, v in cam_dict.items():
                h = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                h.copy_(v, non_blocking=True)
                host_cams[k] = h
        ev = torch.cuda.Event()
        ev.record()
        return img_name, host_pred, host_cams, ev

    def finish(item):
        img_name, host_pred, host_cams, ev = item
        ev.synchronize()
        if host_cams is not None:
            np.save(os.path.join(args.out_cam, img_name + '.npy'), {k: v.numpy() for k, v in host_cams.items()})
        if args.out_cam_pred is not None:
            PIL.Image.fromarray(host_pred.numpy()).save(os.path.join(args.out_cam_pred, img_name + '.png'))

    for d in (args.out_cam, args.out_cam_pred):
        if d is not None:
            os.makedirs(d, exist_ok=True)
    pending = None
    for it, (img_name, img_list, label) in enumerate(loader):
        img_name, label = img_name[0], label[0]
        with PIL.Image.open(wdata.get_img_path(img_name, args.voc12_root)) as im:      # (header only: the reference decodes the image again for its shape)
            W, H = im.size
        norm_cam, pred, cam_dict = infer_image(model, img_list, label, (H, W), args.out_cam_pred_alpha)
        item = start(img_name, pred, cam_dict)
        if pending is not None:
            finish(pending)
        pending = item
    if pending is not None:
        finish(pending)


if __name__ == '__main__':
    main()
