"""Multi-scale CAM inference of contrast_infer.py:58-99 on the HIP kernels.

8 inputs (4 scales x {orig, h-flip}; each pair is one batch-of-two forward); output #2 of the Net (the PCM-refined CAM); bilinear resize to
the original size (align_corners=False), label gating, un-flip and the sum are ONE accumulate kernel per
forward; clamp / per-class min-max normalise / argmax against the bg score are fused in `infer_finish`.
"""
import torch

from . import _lib as L


@torch.no_grad()
def infer_image(model, img_list, label20, orig_size, alpha=0.26):
    """img_list: 8 float tensors [1,3,h,w] (or [3,h,w]) in VOC12ClsDatasetMSF order; label20 [20];
    returns (norm_cam [20,H,W] f32, pred [H,W] uint8, cam_dict {class: map}) — device tensors."""
    dev = next(model.parameters()).device
    H, W = orig_size
    lab = label20.to(dev, non_blocking=True).float().contiguous()
    sum_cam = torch.zeros(20, H, W, device=dev, dtype=torch.float32)
    imgs = []
    for img in img_list:
        img = torch.as_tensor(img).to(dev, non_blocking=True).float()     # (pinned loader batches: asynchronous)
        imgs.append(img.unsqueeze(0) if img.dim() == 3 else img)
    # an image and its flipped copy (same size, consecutive in the MSF order) go through the net as ONE batch of two: every op
    # of the eval forward is per image, so the maps are those of two separate forwards (contrast_infer.py:58-66 runs 8)
    batches, i = [], 0
    while i < len(imgs):
        pair = i + 1 < len(imgs) and imgs[i + 1].shape == imgs[i].shape and imgs[i].shape[0] == 1
        batches.append((i, torch.cat(imgs[i:i + 2]).contiguous() if pair else imgs[i].contiguous()))
        i += 2 if pair else 1
    # ... and two such batches of DIFFERENT sizes are the two row segments of one launch sequence (the engine's two-view form,
    # as the training step batches its 448x448 and 128x128 views): largest with smallest, so both sequences fill the chip
    eng = getattr(model, "_engine", None)
    order = sorted(range(len(batches)), key=lambda k: batches[k][1].shape[2] * batches[k][1].shape[3])
    jobs, maps = [], {}
    while order:
        a = order.pop()
        b = next((k for k in order if batches[k][1].shape[0] == batches[a][1].shape[0]), None) if eng is not None else None
        if b is not None:
            order.remove(b)
        jobs.append((a, b))
    for a, b in jobs:
        if b is None:
            outs = [model(batches[a][1])]
        else:
            outs, _ = eng.active(dev).run_forward([batches[a][1].float(), batches[b][1].float()], save=False)
        for k, out in zip((a, b), outs):
            for j in range(out[1].shape[0]):
                maps[batches[k][0] + j] = out[1][j, 1:].contiguous()      # planes 1..20 (contrast_infer.py:62 `cam[:, 1:, :, :]`, `[0]`)
    for idx in sorted(maps):                                 # accumulate in the reference's order (same float sums)
        m = maps[idx]
        L.resize_planar_fwd(m, sum_cam, 20, m.shape[1], m.shape[2], H, W, False, plane_mul=lab, flip_x=(idx % 2 == 1), accumulate=True)
    stats = torch.empty(20, 6, device=dev, dtype=torch.float32)
    L.plane_stats(sum_cam, stats, 20, H * W)
    norm_cam = torch.empty(20, H, W, device=dev, dtype=torch.float32)
    pred = torch.empty(H, W, device=dev, dtype=torch.uint8)
    L.infer_finish(sum_cam, stats, alpha, norm_cam, pred, H * W)
    cam_dict = {i: norm_cam[i] for i in range(20) if float(label20[i]) > 1e-5}
    return norm_cam, pred, cam_dict
