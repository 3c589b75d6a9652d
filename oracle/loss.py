"""CPU restatement of the contrast_train.py loss step (test infrastructure, see oracle/__init__.py).

Follows the body of the training loop, /root/reference/contrast_train.py:129-395, written as
functions.  The Python `random` stream the reference consumes (never seeded by the script) is an
explicit `rng` argument (`random.Random(seed)` reproduces `random.seed(seed)` exactly).
"""
import random as _random

import numpy as np
import torch
import torch.nn.functional as F

from . import net as onet

TAU = 0.1


def adaptive_min_pooling_loss(x):
    """contrast_train.py:16-25."""
    n, c, h, w = x.size()
    k = h * w // 4
    x = torch.max(x, dim=1)[0]
    y = torch.topk(x.view(n, -1), k=k, dim=-1, largest=False)[0]
    y = F.relu(y)
    return torch.sum(y) / (k * n)


def max_onehot(x):
    """contrast_train.py:28-32 (in place on the tensor it is handed, like the reference)."""
    x_max = torch.max(x[:, 1:, :, :], dim=1, keepdim=True)[0]
    x[:, 1:, :, :][x[:, 1:, :, :] != x_max] = 0
    return x


def max_norm(p, e=1e-5):
    """tool/visualization.py:62-67 (4-D torch branch); min/max carry gradient."""
    N, C, H, W = p.size()
    p = F.relu(p)
    max_v = torch.max(p.view(N, C, -1), dim=-1)[0].view(N, C, 1, 1)
    min_v = torch.min(p.view(N, C, -1), dim=-1)[0].view(N, C, 1, 1)
    return F.relu(p - min_v - e) / (max_v - min_v + e)


def _resize(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


def pseudo_labels_and_prototypes(cam_rv_down, f_proj, label, bg_threshold):
    """contrast_train.py:184-209 (view 1) == :212-241 (view 2); all under no_grad.
    cam_rv_down [N,21,h,w] (detached here), f_proj [N,128,h,w].
    Returns (pseudo_label [P] int64, prototypes [21,128], normalised cam [N,21,h,w])."""
    with torch.no_grad():
        fea = f_proj.detach()
        c_fea = fea.shape[1]
        cam = F.relu(cam_rv_down.detach())
        n, c, h, w = cam.shape
        mx = torch.max(cam.view(n, c, -1), dim=-1)[0].view(n, c, 1, 1)
        mn = torch.min(cam.view(n, c, -1), dim=-1)[0].view(n, c, 1, 1)
        cam[cam < mn + 1e-5] = 0.
        cam = (cam - mn - 1e-5) / (mx - mn + 1e-5)
        cam[:, 0, :, :] = bg_threshold
        scores = F.softmax(cam * label, dim=1)
        pseudo = scores.argmax(dim=1, keepdim=True)
        fea = fea.permute(0, 2, 3, 1).reshape(-1, c_fea)
        top_values, top_indices = torch.topk(cam.transpose(0, 1).reshape(c, -1), k=h * w // 8, dim=-1)
        protos = torch.zeros(c, c_fea)
        for i in range(c):
            top_fea = fea[top_indices[i]]
            protos[i] = torch.sum(top_values[i].unsqueeze(-1) * top_fea, dim=0) / torch.sum(top_values[i])
        protos = F.normalize(protos, dim=-1)
    return pseudo.reshape(-1), protos, cam


def nce(f, pos, protos):
    """-mean log( exp(f.pos/tau) / sum_c exp(f.proto_c/tau) ) — contrast_train.py:261-263."""
    a = torch.exp(torch.sum(f * pos, dim=-1) / TAU)
    b = torch.sum(torch.exp(torch.matmul(f, protos.transpose(0, 1)) / TAU), dim=-1)
    return torch.mean(-1 * torch.log(a / b))


def intra_view_nce(f, pseudo, protos, rng, record=None):
    """contrast_train.py:285-334 (== :338-387).  f [P,128] L2-normalised (with grad),
    pseudo [P], protos [21,128]."""
    P = f.shape[0]
    pos = protos[pseudo]
    sim = (torch.sum(f * pos, dim=-1) + 1) / 2.
    a1 = torch.exp(torch.sum(f * pos, dim=-1) / TAU)
    neg_scores = torch.matmul(f, protos.transpose(0, 1))
    # :291 — P draws of 10-of-21 whose result is never used, but which advance the RNG
    for _ in range(P):
        rng.sample(range(21), 10)
    with torch.no_grad():
        _, lower = torch.topk(neg_scores, k=13, largest=True, dim=-1)
        lower = lower[:, 3:]
    negs = protos.unsqueeze(0).repeat(P, 1, 1)
    lower_negs = negs[torch.arange(P).unsqueeze(1), lower]
    cand = torch.cat([pos.unsqueeze(1), lower_negs], dim=1)            # [P,11,128]
    a2 = torch.sum(torch.exp(torch.matmul(f.unsqueeze(1), cand.transpose(1, 2)).squeeze(1) / TAU), dim=-1)
    loss = torch.zeros(1)
    C = 0
    exists = np.unique(pseudo.numpy()).tolist()
    for cls in range(21):
        if cls not in exists:
            continue
        C += 1
        sel = pseudo == cls
        a1c, a2c, simc = a1[sel], a2[sel], sim[sel]
        n_c = a1c.shape[0]
        if n_c < 2:
            continue
        ridx = torch.tensor(rng.sample(range(n_c), n_c // 2)).long()
        with torch.no_grad():
            k = int(n_c * 0.6)
            _, lidx = torch.topk(simc, k=k, largest=False)
            lidx = lidx[k - n_c // 2:]
        if record is not None:
            record.append((cls, ridx.clone(), lidx.clone()))
        a1s = torch.cat([a1c[ridx], a1c[lidx]], dim=0).reshape(-1)
        a2s = torch.cat([a2c[ridx], a2c[lidx]], dim=0).reshape(-1)
        loss = loss + torch.mean(-1 * torch.log(a1s / a2s))
    return loss / C


def step_loss(out1, out2, label20, bg_threshold=0.20, rng=None, extras=None, inject=None):
    """contrast_train.py:138-395 given the two forward 4-tuples.
    out_v = (cam_v, cam_rv_v, f_proj_v, cam_rv_v_down); label20 float [N,20].
    Returns dict of the 8 logged scalars (tensors; 'loss' carries the graph).
    inject: optional {protos1, protos2, pseudo1, pseudo2} — ANOTHER implementation's discrete selections (its per-class top-32 prototypes, its
    pseudo-labels) used in place of this one's: a near-tie at the top-32 boundary resolved differently moves a prototype by ~4e-3 and the NCE terms
    by ~1e-4 at P = 512, without any arithmetic difference (tests/test_gpu_loss.py)."""
    rng = rng if rng is not None else _random.Random(0)
    cam1, cam_rv1, f_proj1, cam_rv1_down = out1
    cam2, cam_rv2, f_proj2, cam_rv2_down = out2
    N = cam1.shape[0]
    label = torch.cat((torch.ones((N, 1)), label20), dim=1).unsqueeze(2).unsqueeze(3)

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

    cam1[:, 0, :, :] = 1 - torch.max(cam1[:, 1:, :, :], dim=1)[0]
    cam2[:, 0, :, :] = 1 - torch.max(cam2[:, 1:, :, :], dim=1)[0]
    k_ecr = int(21 * hs * ws * 0.2)
    tensor_ecr1 = torch.abs(max_onehot(cam2.detach()) - cam_rv1)
    tensor_ecr2 = torch.abs(max_onehot(cam1.detach()) - cam_rv2)
    loss_ecr1 = torch.mean(torch.topk(tensor_ecr1.view(ns, -1), k=k_ecr, dim=-1)[0])
    loss_ecr2 = torch.mean(torch.topk(tensor_ecr2.view(ns, -1), k=k_ecr, dim=-1)[0])
    loss_ecr = loss_ecr1 + loss_ecr2
    loss_cls = (loss_cls1 + loss_cls2) / 2 + (loss_rvmin1 + loss_rvmin2) / 2

    # ---- contrastive part (:179-387)
    f_proj1 = _resize(f_proj1, (16, 16))
    cam_rv1_down = _resize(cam_rv1_down, (16, 16))
    pseudo1, protos1, ncam1 = pseudo_labels_and_prototypes(cam_rv1_down, f_proj1, label, bg_threshold)
    pseudo2, protos2, ncam2 = pseudo_labels_and_prototypes(cam_rv2_down, f_proj2, label, bg_threshold)

    if inject is not None:
        pseudo1, pseudo2 = inject.get("pseudo1", pseudo1), inject.get("pseudo2", pseudo2)
        protos1, protos2 = inject.get("protos1", protos1), inject.get("protos2", protos2)

    def rows(fp):
        n_f, c_f, h_f, w_f = fp.shape
        return F.normalize(fp.permute(0, 2, 3, 1).reshape(n_f * h_f * w_f, c_f), dim=-1)

    f1, f2 = rows(f_proj1), rows(f_proj2)
    loss_cross_nce = TAU * (nce(f1, protos2[pseudo1], protos2) + nce(f2, protos1[pseudo2], protos1)) / 2
    loss_cross_nce2 = TAU * (nce(f1, protos1[pseudo2], protos1) + nce(f2, protos2[pseudo1], protos2)) / 2
    rec1, rec2 = [], []
    intra1 = intra_view_nce(f1, pseudo1, protos1, rng, rec1)
    intra2 = intra_view_nce(f2, pseudo2, protos2, rng, rec2)
    loss_intra_nce = TAU * (intra1 + intra2) / 2
    loss_nce = loss_cross_nce + loss_cross_nce2 + loss_intra_nce
    loss = loss_cls + loss_er + loss_ecr + loss_nce
    if extras is not None:
        extras.update(pseudo1=pseudo1, pseudo2=pseudo2, protos1=protos1, protos2=protos2,
                      f1=f1, f2=f2, sel1=rec1, sel2=rec2, ncam1=ncam1, ncam2=ncam2,
                      loss_cls1=loss_cls1, loss_cls2=loss_cls2, loss_rvmin1=loss_rvmin1,
                      loss_rvmin2=loss_rvmin2, loss_ecr1=loss_ecr1, loss_ecr2=loss_ecr2,
                      intra1=intra1, intra2=intra2)
    return dict(loss=loss.reshape(()), loss_cls=loss_cls, loss_er=loss_er, loss_ecr=loss_ecr,
                loss_nce=loss_nce.reshape(()), loss_intra_nce=loss_intra_nce.reshape(()),
                loss_cross_nce=loss_cross_nce, loss_cross_nce2=loss_cross_nce2)


def train_step(img1, label20, sd, masks1=None, masks2=None, bg_threshold=0.20, rng=None, extras=None, gates1=None, gates2=None, inject=None):
    """One loop body, contrast_train.py:130-395: second view, two forwards, the loss.  gates1 / gates2: optional injected ReLU
    decisions per view (oracle/net.py `_relu`)."""
    img2 = F.interpolate(img1, size=(128, 128), mode="bilinear", align_corners=True)
    out1 = onet.net_forward(img1, sd, masks1, gates=gates1)
    out2 = onet.net_forward(img2, sd, masks2, gates=gates2)
    if extras is not None:
        extras.update(out1=out1, out2=out2)
    return step_loss(out1, out2, label20, bg_threshold, rng, extras, inject)
