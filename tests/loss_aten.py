"""Loss step of contrast_train.py:138-395 as device-side torch ops (STAGING PATH).

This is how a user of the reference's own training script would drive the drop-in Net: the loss
body stays PyTorch, on GPU tensors, behind the `Net.forward` 4-tuple.  wseg_amd/loss_hip.py replaces
it kernel by kernel (the 448^2 maps are never materialised there); this module stays as the
on-device cross-check for those kernels.  It never runs on the CPU in the product path
(train.Trainer refuses non-GPU tensors).

Randomness (contrast_train.py:291,316,344,370 use the never-seeded Python `random`):
  rng_parity=True  — consume a `random.Random` exactly like the reference, including the 2*P
                     discarded `sample(range(21), 10)` draws (needed for bit-parity tests; ~35 ms of
                     host time per step at P=4096);
  rng_parity=False — same sampling distribution from the supplied `random.Random`, without the
                     discarded draws.
"""
import random as _random

import torch
import torch.nn.functional as F

TAU = 0.1


def adaptive_min_pooling_loss(x):                                   # contrast_train.py:16-25
    n, c, h, w = x.size()
    k = h * w // 4
    x = torch.max(x, dim=1)[0]
    y = torch.topk(x.view(n, -1), k=k, dim=-1, largest=False)[0]
    return torch.sum(F.relu(y)) / (k * n)


def max_onehot(x):                                                  # contrast_train.py:28-32
    x_max = torch.max(x[:, 1:, :, :], dim=1, keepdim=True)[0]
    x[:, 1:, :, :][x[:, 1:, :, :] != x_max] = 0
    return x


def max_norm(p, e=1e-5):                                            # tool/visualization.py:62-67
    N, C, H, W = p.size()
    p = F.relu(p)
    max_v = torch.max(p.view(N, C, -1), dim=-1)[0].view(N, C, 1, 1)
    min_v = torch.min(p.view(N, C, -1), dim=-1)[0].view(N, C, 1, 1)
    return F.relu(p - min_v - e) / (max_v - min_v + e)


def _resize(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


def pseudo_labels_and_prototypes(cam_rv_down, f_proj, label, bg_threshold, bg_topk_idx=None):
    """contrast_train.py:184-209 / :212-241 (no_grad)."""
    with torch.no_grad():
        fea = f_proj.detach()
        c_fea = fea.shape[1]
        cam = F.relu(cam_rv_down.detach())
        n, c, h, w = cam.shape
        mx = torch.max(cam.view(n, c, -1), dim=-1)[0].view(n, c, 1, 1)
        mn = torch.min(cam.view(n, c, -1), dim=-1)[0].view(n, c, 1, 1)
        cam = torch.where(cam < mn + 1e-5, torch.zeros_like(cam), cam)
        cam = (cam - mn - 1e-5) / (mx - mn + 1e-5)
        cam[:, 0, :, :] = bg_threshold
        pseudo = F.softmax(cam * label, dim=1).argmax(dim=1).reshape(-1)
        fea = fea.permute(0, 2, 3, 1).reshape(-1, c_fea)
        rows_t = cam.transpose(0, 1).reshape(c, -1)
        top_values, top_indices = torch.topk(rows_t, k=h * w // 8, dim=-1)
        if bg_topk_idx is not None:
            # Q5: torch.topk's pick among exactly-tied values is implementation-defined.  The bg row (constant
            # bg_threshold) and every class that never wins the CAM gate in the batch (constant -1 row) are
            # fully tied; use the CPU library's data-independent index set for them.
            const = rows_t.max(dim=1)[0] == rows_t.min(dim=1)[0]
            top_indices[const] = bg_topk_idx.to(device=top_indices.device, dtype=top_indices.dtype)
        top_fea = fea[top_indices]                                    # [21,32,128]
        protos = (top_values.unsqueeze(-1) * top_fea).sum(1) / top_values.sum(1, keepdim=True)
        protos = F.normalize(protos, dim=-1)
    return pseudo, protos


def _nce(f, pos, protos):                                           # contrast_train.py:261-263
    a = torch.exp(torch.sum(f * pos, dim=-1) / TAU)
    b = torch.sum(torch.exp(torch.matmul(f, protos.t()) / TAU), dim=-1)
    return torch.mean(-torch.log(a / b))


def _intra(f, pseudo, protos, rng, rng_parity):
    """contrast_train.py:285-334."""
    P = f.shape[0]
    pos = protos[pseudo]
    dot = torch.sum(f * pos, dim=-1)
    sim = (dot + 1) / 2.
    a1 = torch.exp(dot / TAU)
    scores = torch.matmul(f, protos.t())
    if rng_parity:
        for _ in range(P):
            rng.sample(range(21), 10)
    with torch.no_grad():
        lower = torch.topk(scores, k=13, largest=True, dim=-1)[1][:, 3:]
    a2 = a1 + torch.exp(torch.gather(scores, 1, lower) / TAU).sum(-1)
    labels_host = pseudo.cpu()
    loss = f.new_zeros(())
    C = 0
    for cls in torch.unique(labels_host).tolist():
        C += 1
        idx = (labels_host == cls).nonzero(as_tuple=True)[0].to(f.device)
        n_c = idx.numel()
        if n_c < 2:
            continue
        ridx = torch.tensor(rng.sample(range(n_c), n_c // 2), device=f.device, dtype=torch.long)
        with torch.no_grad():
            k = int(n_c * 0.6)
            lidx = torch.topk(sim[idx], k=k, largest=False)[1][k - n_c // 2:]
        sel = torch.cat([idx[ridx], idx[lidx]])
        loss = loss + torch.mean(-torch.log(a1[sel] / a2[sel]))
    return loss / C


def step_loss(out1, out2, label20, bg_threshold=0.20, rng=None, rng_parity=False, bg_topk_idx=None):
    """out_v = (cam_v, cam_rv_v, f_proj_v, cam_rv_v_down) on the GPU; returns the 8 logged scalars."""
    rng = rng if rng is not None else _random
    cam1, cam_rv1, f_proj1, cam_rv1_down = out1
    cam2, cam_rv2, f_proj2, cam_rv2_down = out2
    N = cam1.shape[0]
    dev = cam1.device
    label = torch.cat((torch.ones((N, 1), device=dev), label20.to(dev)), dim=1).unsqueeze(2).unsqueeze(3)

    label1 = F.adaptive_avg_pool2d(cam1, (1, 1))
    loss_rvmin1 = adaptive_min_pooling_loss((cam_rv1 * label)[:, 1:, :, :])
    cam1 = _resize(max_norm(cam1), (128, 128)) * label
    cam_rv1 = _resize(max_norm(cam_rv1), (128, 128)) * label
    label2 = F.adaptive_avg_pool2d(cam2, (1, 1))
    loss_rvmin2 = adaptive_min_pooling_loss((cam_rv2 * label)[:, 1:, :, :])
    cam2 = max_norm(cam2) * label
    cam_rv2 = max_norm(cam_rv2) * label
    loss_cls1 = F.multilabel_soft_margin_loss(label1[:, 1:, :, :], label[:, 1:, :, :])
    loss_cls2 = F.multilabel_soft_margin_loss(label2[:, 1:, :, :], label[:, 1:, :, :])
    ns, cs, hs, ws = cam2.size()
    loss_er = torch.mean(torch.abs(cam1[:, 1:, :, :] - cam2[:, 1:, :, :]))
    cam1 = torch.cat([1 - torch.max(cam1[:, 1:], dim=1, keepdim=True)[0], cam1[:, 1:]], dim=1)
    cam2 = torch.cat([1 - torch.max(cam2[:, 1:], dim=1, keepdim=True)[0], cam2[:, 1:]], dim=1)
    k_ecr = int(21 * hs * ws * 0.2)
    t1 = torch.abs(max_onehot(cam2.detach().clone()) - cam_rv1)
    t2 = torch.abs(max_onehot(cam1.detach().clone()) - cam_rv2)
    loss_ecr = torch.mean(torch.topk(t1.view(ns, -1), k=k_ecr, dim=-1)[0]) + \
        torch.mean(torch.topk(t2.view(ns, -1), k=k_ecr, dim=-1)[0])
    loss_cls = (loss_cls1 + loss_cls2) / 2 + (loss_rvmin1 + loss_rvmin2) / 2

    f_proj1 = _resize(f_proj1, (16, 16))
    cam_rv1_down = _resize(cam_rv1_down, (16, 16))
    pseudo1, protos1 = pseudo_labels_and_prototypes(cam_rv1_down, f_proj1, label, bg_threshold, bg_topk_idx)
    pseudo2, protos2 = pseudo_labels_and_prototypes(cam_rv2_down, f_proj2, label, bg_threshold, bg_topk_idx)

    def rows(fp):
        n_f, c_f, h_f, w_f = fp.shape
        return F.normalize(fp.permute(0, 2, 3, 1).reshape(n_f * h_f * w_f, c_f), dim=-1)

    f1, f2 = rows(f_proj1), rows(f_proj2)
    loss_cross_nce = TAU * (_nce(f1, protos2[pseudo1], protos2) + _nce(f2, protos1[pseudo2], protos1)) / 2
    loss_cross_nce2 = TAU * (_nce(f1, protos1[pseudo2], protos1) + _nce(f2, protos2[pseudo1], protos2)) / 2
    intra1 = _intra(f1, pseudo1, protos1, rng, rng_parity)
    intra2 = _intra(f2, pseudo2, protos2, rng, rng_parity)
    loss_intra_nce = TAU * (intra1 + intra2) / 2
    loss_nce = loss_cross_nce + loss_cross_nce2 + loss_intra_nce
    loss = loss_cls + loss_er + loss_ecr + loss_nce
    return dict(loss=loss, loss_cls=loss_cls, loss_er=loss_er, loss_ecr=loss_ecr, loss_nce=loss_nce,
                loss_intra_nce=loss_intra_nce, loss_cross_nce=loss_cross_nce, loss_cross_nce2=loss_cross_nce2)
