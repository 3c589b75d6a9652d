#!/usr/bin/env python
"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF (build container only).

  python oracle/make_goldens.py [--out tests/golden]

Imports /root/reference (network.resnet38_contrast.Net, tool.torchutils.PolyOptimizer,
tool.visualization.max_norm) and exec()s the reference's own loop-body text
(contrast_train.py:129-395 with the hard `.cuda()` calls removed — the recipe of SURVEY.md §8c)
on procedural weights / inputs from wseg_amd.synth.  Nothing of the reference is copied into
the repo: only inputs' seeds and the numeric outputs are stored.  Dropout2d masks are injected
through forward hooks so the reference and the restatement see identical masks.
"""
import argparse
import os
import random
import sys
import textwrap
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from wseg_amd import synth  # noqa: E402


def _stub_absent_modules():
    """cv2 / tensorboardX / torchvision / pydensecrf / imageio are imported at module top by
    the reference but unused by the functions needed here."""
    for name in ["cv2", "tensorboardX", "torchvision", "torchvision.transforms", "pydensecrf",
                 "pydensecrf.densecrf", "pydensecrf.utils", "imageio"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                if name == "tensorboardX":
                    m.SummaryWriter = object
                if name == "pydensecrf.utils":
                    m.unary_from_softmax = None
                sys.modules[name] = m


def load_reference():
    sys.path.insert(0, REF)
    _stub_absent_modules()
    import network.resnet38_contrast as R
    from tool import visualization, torchutils
    return R, visualization, torchutils


def install_masks(model, mask_sets):
    """Forward hooks that replace each Dropout2d output by input * injected mask.
    mask_sets: list (one per forward call) of dicts from synth.synthetic_dropout_masks."""
    state = {"call": 0}
    sites = {"b6.dropout_2b1": model.b6.dropout_2b1, "b6.dropout_2b2": model.b6.dropout_2b2,
             "b7.dropout_2b1": model.b7.dropout_2b1, "b7.dropout_2b2": model.b7.dropout_2b2,
             "dropout7": model.dropout7}
    handles = []
    for key, mod in sites.items():
        def hook(mod_, inp, out, key=key):
            m = mask_sets[state["call"]][key]
            return inp[0] * m.view(m.shape[0], m.shape[1], 1, 1)
        handles.append(mod.register_forward_hook(hook))
    # advance the mask set after each full forward (dropout7 is the last site)
    def adv(mod_, inp, out):
        state["call"] += 1
    handles.append(model.register_forward_hook(adv))
    return handles


def body_source():
    lines = open(os.path.join(REF, "contrast_train.py")).read().split("\n")
    helpers = "\n".join(lines[15:32])                      # :16-32 adaptive_min_pooling_loss, max_onehot
    body = textwrap.dedent("\n".join(lines[128:395]))      # :129-395
    body = body.replace(".cuda(non_blocking=True)", "").replace(".cuda()", "")
    return helpers, body


def slc(t, step=8):
    return t.detach()[..., ::step, ::step].contiguous().numpy()


def fwd_golden(R, out_dir, name, n, size, seed, sd):
    model = R.Net()
    model.load_state_dict(sd)
    model.eval()
    x = synth.synthetic_images(n, size, seed)
    with torch.no_grad():
        cam, cam_rv, f_proj, cam_rv_down = model(x)
    np.savez_compressed(
        os.path.join(out_dir, name + ".npz"), n=n, size=np.array(size), seed=seed,
        cam_s=slc(cam, 4), cam_rv_s=slc(cam_rv, 4), f_proj=f_proj.numpy(), cam_rv_down=cam_rv_down.numpy(),
        cam_argmax=cam.argmax(1).numpy().astype(np.uint8), cam_rv_argmax=cam_rv.argmax(1).numpy().astype(np.uint8),
        sums=np.array([cam.double().sum().item(), cam_rv.double().sum().item(),
                       f_proj.double().sum().item(), cam_rv_down.double().sum().item()]))
    print("wrote", name, [tuple(t.shape) for t in (cam, cam_rv, f_proj, cam_rv_down)])


GRAD_KEYS = ["fc8.weight", "fc_proj.weight", "f9.weight", "f8_3.weight", "f8_4.weight",
             "b7.conv_branch2b1.weight", "b7.conv_branch1.weight", "b6.conv_branch2a.weight",
             "b5.conv_branch2a.weight", "b4.conv_branch2a.weight", "b4.conv_branch1.weight",
             "b3.conv_branch2a.weight", "b3_1.conv_branch2b1.weight"]


def step_golden(R, visualization, out_dir, name, n, size, seed, sd, py_seed, edge_labels=False):
    model = R.Net()
    model.load_state_dict(sd)
    model.train()
    masks = [synth.synthetic_dropout_masks(n, seed * 2 + 0), synth.synthetic_dropout_masks(n, seed * 2 + 1)]
    install_masks(model, masks)
    img = synth.synthetic_images(n, size, seed)
    lab = synth.synthetic_labels(n, seed)
    if edge_labels:                                  # an image without any foreground class, one with all twenty
        lab = lab.clone(); lab[0] = 0.0; lab[1] = 1.0
    helpers, body = body_source()
    ns = {"torch": torch, "F": F, "np": np, "random": random, "visualization": visualization,
          "model": model, "pack": (None, img, lab), "args": types.SimpleNamespace(bg_threshold=0.20)}
    exec(helpers, ns)
    random.seed(py_seed)
    exec(body, ns)
    loss = ns["loss"]
    loss.backward()
    scal = {k: float(ns[k]) for k in ["loss", "loss_cls", "loss_er", "loss_ecr", "loss_nce",
                                      "loss_intra_nce", "loss_cross_nce", "loss_cross_nce2"]}
    params = dict(model.named_parameters())
    grads = {}
    n_with_grad = 0
    for k, p in params.items():
        if p.grad is not None:
            n_with_grad += 1
    for k in GRAD_KEYS:
        g = params[k].grad
        flat = g.reshape(-1)
        grads["gnorm/" + k] = np.array(g.double().norm().item())
        grads["gsum/" + k] = np.array(g.double().sum().item())
        step = max(1, flat.numel() // 4096)
        grads["gslice/" + k] = flat[::step][:4096].numpy().copy()
    np.savez_compressed(
        os.path.join(out_dir, name + ".npz"), n=n, size=size, seed=seed, py_seed=py_seed,
        n_with_grad=n_with_grad, label=lab.numpy(),
        **{"s/" + k: np.array(v) for k, v in scal.items()},
        protos1=ns["prototypes1"].numpy(), protos2=ns["prototypes2"].numpy(),
        pseudo1=ns["pseudo_label1"].numpy().astype(np.uint8), pseudo2=ns["pseudo_label2"].numpy().astype(np.uint8),
        f1_s=ns["f_proj1"].detach().numpy()[::7].copy(), f2_s=ns["f_proj2"].detach().numpy()[::7].copy(),
        **grads)
    print("wrote", name, scal, "params with grad:", n_with_grad)


def _run_body(R, visualization, n, size, seed, sd, py_seed, bf16=False):
    """One pass of the reference's loop-body text (contrast_train.py:129-395) + backward on the reference Net; returns its namespace and the model.
    bf16=True: the reference ITSELF with weights and activations in bfloat16 (`model.bfloat16()`, the images cast on entry, the four outputs cast back)
    and the loss maths in float32 — what the reference shows in the storage precision of the benchmarked mode."""
    model = R.Net()
    model.load_state_dict(sd)
    model.train()
    masks = [synth.synthetic_dropout_masks(n, seed * 2 + 0), synth.synthetic_dropout_masks(n, seed * 2 + 1)]
    if bf16:
        model.bfloat16()
        masks = [{k: v.bfloat16() for k, v in m.items()} for m in masks]
    install_masks(model, masks)
    img = synth.synthetic_images(n, size, seed)
    lab = synth.synthetic_labels(n, seed)
    helpers, body = body_source()
    call = (lambda x: tuple(o.float() for o in model(x.bfloat16()))) if bf16 else model
    ns = {"torch": torch, "F": F, "np": np, "random": random, "visualization": visualization,
          "model": call, "pack": (None, img, lab), "args": types.SimpleNamespace(bg_threshold=0.20)}
    exec(helpers, ns)
    random.seed(py_seed)
    exec(body, ns)
    ns["loss"].backward()
    return ns, model


SCALAR_KEYS = ["loss", "loss_cls", "loss_er", "loss_ecr", "loss_nce", "loss_intra_nce", "loss_cross_nce", "loss_cross_nce2"]


def step_refbf16_golden(R, visualization, out_dir, name, n, size, seed, sd, py_seed):
    """<name>_refbf16.npz: the 8 scalars, the 13 gradient slices / norms, pseudo-labels and prototypes of the REFERENCE run in bfloat16
    (see _run_body) on the inputs of fixture <name>.  The GPU tests hold the HIP bf16 mode's deviation from the fp32 fixture to a
    multiple of THIS run's deviation from it (tests/test_gpu_loss.py `test_bf16_within_the_reference_bf16_envelope`)."""
    ns, model = _run_body(R, visualization, n, size, seed, sd, py_seed, bf16=True)
    scal = {k: float(ns[k]) for k in SCALAR_KEYS}
    params = dict(model.named_parameters())
    grads = {}
    for k in GRAD_KEYS:
        g = params[k].grad.float()
        flat = g.reshape(-1)
        grads["gnorm/" + k] = np.array(g.double().norm().item())
        step = max(1, flat.numel() // 4096)
        grads["gslice/" + k] = flat[::step][:4096].numpy().copy()
    np.savez_compressed(
        os.path.join(out_dir, name + "_refbf16.npz"), n=n, size=size, seed=seed, py_seed=py_seed,
        **{"s/" + k: np.array(v) for k, v in scal.items()},
        protos1=ns["prototypes1"].float().numpy(), protos2=ns["prototypes2"].float().numpy(),
        pseudo1=ns["pseudo_label1"].numpy().astype(np.uint8), pseudo2=ns["pseudo_label2"].numpy().astype(np.uint8), **grads)
    print("wrote", name + "_refbf16", scal)


def step_margins_golden(R, visualization, out_dir, name, n, size, seed, sd, py_seed):
    """<name>_margins.npz: the reference's OWN near-tie evidence for fixture <name> — per class and view the 33 largest values of the
    prototype top-k input (contrast_train.py:202-203: the 32nd / 33rd gap is the selection boundary) with the 32 selected pixel indices, and per
    contrast pixel the top-1 / top-2 gap of the pseudo-label scores (:195-197).  A differing prototype member or pseudo-label is then checked against
    the REFERENCE's margin at that very class / pixel."""
    ns, model = _run_body(R, visualization, n, size, seed, sd, py_seed)
    out = {}
    for v, (tv, ti) in (("1", ("top_values", "top_indices")), ("2", ("top_values2", "top_indices2"))):
        rows = ns["cam_rv%s_down" % v].transpose(0, 1).reshape(21, -1)
        t33 = torch.topk(rows, 33, dim=-1)[0]
        assert torch.equal(t33[:, :32], ns[tv])
        out["top33_" + v] = t33.numpy()
        out["topidx_" + v] = ns[ti].numpy().astype(np.int32)
        sc = ns["scores" + v]                                        # [21, n, h, w] after the transpose of :199
        s2 = torch.topk(sc.reshape(21, -1), 2, dim=0)[0]
        out["label_margin_" + v] = (s2[0] - s2[1]).numpy().astype(np.float32)
    for k in SCALAR_KEYS:                                            # (the same run as the fixture: must reproduce its scalars)
        out["s/" + k] = np.array(float(ns[k]))
    np.savez_compressed(os.path.join(out_dir, name + "_margins.npz"), **out)
    print("wrote", name + "_margins", {k: float(np.min(out["top33_" + k][1:, 31] - out["top33_" + k][1:, 32])) for k in "12"})


def sgd_golden(torchutils, out_dir):
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(5)
    ps = [torch.randn(7, 5, generator=g), torch.randn(11, generator=g), torch.randn(3, 4, generator=g)]
    params = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torchutils.PolyOptimizer([
        {"params": [params[0]], "lr": 0.01, "weight_decay": 5e-4},
        {"params": [params[1]], "lr": 0.02, "weight_decay": 0},
        {"params": [params[2]], "lr": 0.1, "weight_decay": 5e-4},
        {"params": [], "lr": 0.2, "weight_decay": 0}], lr=0.01, weight_decay=5e-4, max_step=10)
    grads = [[torch.randn(p.shape, generator=g) for p in ps] for _ in range(3)]
    for s in range(3):
        for p, gr in zip(params, grads[s]):
            p.grad = gr.clone()
        if s == 1:
            params[1].grad = None                  # a skipped parameter
        opt.step()
    out = {"momentum": np.array(opt.param_groups[0]["momentum"]), "lr_final": np.array([g_["lr"] for g_ in opt.param_groups])}
    for i in range(3):
        out[f"p{i}_init"] = ps[i].numpy()
        out[f"p{i}_final"] = params[i].detach().numpy()
        for s in range(3):
            out[f"g{s}_{i}"] = grads[s][i].numpy()
    np.savez_compressed(os.path.join(out_dir, "sgd_3steps.npz"), **out)
    print("wrote sgd_3steps; SGD momentum actually used:", opt.param_groups[0]["momentum"])


def topk_pattern_golden(out_dir):
    out = {}
    for n in (512, 1024, 4096, 32768):
        v, i = torch.topk(torch.full((21, n), 0.2)[0:1], 32, dim=-1)
        out[f"n{n}"] = i[0].numpy()
        t = torch.full((21, n), 0.2)
        t[1:] = torch.rand(20, n)
        out[f"row0_n{n}"] = torch.topk(t, 32, dim=-1)[1][0].numpy()
    np.savez_compressed(os.path.join(out_dir, "bg_topk_pattern.npz"), **out)
    print("wrote bg_topk_pattern", {k: v[:4].tolist() for k, v in out.items()})


def infer_golden(R, out_dir, sd, name="infer_1img", H=40, W=56, classes=(3, 11), seed0=40, store_stride=1, extras=None):
    """contrast_infer.py:49-99 on one synthetic image: the 8 MSF inputs (4 scales x flip, sizes round(H*s) x round(W*s) as
    voc12/data.py:100-121 makes them), the reference Net, and the reference's own post-process text (:75-80, :97-98)."""
    # extras = "margins": write <name>_margins.npz (the reference's per-pixel top-1 / top-2 margin of [alpha, present classes], float16) instead of
    # the fixture; extras = "refbf16": <name>_refbf16.npz, the arg-max map of the reference run in bfloat16 (model.bfloat16(), post-process in f32)
    model = R.Net()
    model.load_state_dict(sd)
    model.eval()
    if extras == "refbf16":
        model.bfloat16()
    lab = torch.zeros(20)
    lab[list(classes)] = 1
    imgs = []
    for si, s in enumerate([0.5, 1.0, 1.5, 2.0]):
        hs, ws = int(np.round(H * s)), int(np.round(W * s))
        im = synth.synthetic_images(1, (hs, ws), seed0 + si)
        imgs += [im, torch.flip(im, dims=[3])]
    cam_list = []
    for i, img in enumerate(imgs):
        with torch.no_grad():                                       # contrast_infer.py:58-66
            _, cam, _, _ = model(img.bfloat16() if extras == "refbf16" else img)
            cam = F.interpolate(cam.float()[:, 1:, :, :], (H, W), mode="bilinear", align_corners=False)[0]
            cam = cam.numpy() * lab.clone().view(20, 1, 1).numpy()
            if i % 2 == 1:
                cam = np.flip(cam, axis=-1)
            cam_list.append(cam)
    lines = open(os.path.join(REF, "contrast_infer.py")).read().split("\n")
    post = textwrap.dedent("\n".join(lines[74:80]))                 # :75-80
    pred_src = textwrap.dedent("\n".join(lines[96:98]))             # :97-98
    ns = {"np": np, "cam_list": cam_list, "args": types.SimpleNamespace(out_cam_pred_alpha=0.26)}
    exec(post, ns)
    exec(pred_src, ns)
    norm_cam = ns["norm_cam"].astype(np.float32)
    if extras == "refbf16":
        np.savez_compressed(os.path.join(out_dir, name + "_refbf16.npz"), H=H, W=W, pred=ns["pred"].astype(np.uint8))
        print("wrote", name + "_refbf16", np.bincount(ns["pred"].reshape(-1)))
        return
    if extras == "margins":
        stack = np.concatenate([np.full((1, H, W), 0.26, np.float32), norm_cam[sorted(classes)]], axis=0)
        top2 = np.sort(stack, axis=0)[-2:]
        np.savez_compressed(os.path.join(out_dir, name + "_margins.npz"), H=H, W=W, classes=np.array(sorted(classes)),
                            margin=(top2[1] - top2[0]).astype(np.float16), pred=ns["pred"].astype(np.uint8))
        print("wrote", name + "_margins", float((top2[1] - top2[0]).min()))
        return
    extra = {}
    if name == "infer_1img":
        extra["norm_cam"] = norm_cam                                 # (round-1 layout: all 20 planes)
    else:                                                            # present planes only (the others are the constant -1), optionally strided
        extra["classes"] = np.array(sorted(classes))
        extra["norm_cam_present"] = norm_cam[sorted(classes)][:, ::store_stride, ::store_stride].copy()
        extra["store_stride"] = store_stride
        extra["absent_value"] = np.array([norm_cam[c].min() for c in range(20) if c not in classes] + [norm_cam[[c for c in range(20) if c not in classes][0]].max()])
        extra["sums"] = norm_cam[sorted(classes)].astype(np.float64).sum(axis=(1, 2))
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), H=H, W=W, label=lab.numpy(), seed0=seed0,
                        pred=ns["pred"].astype(np.uint8), **extra)
    print("wrote", name, norm_cam.shape, np.bincount(ns["pred"].reshape(-1)))


def multistep_golden(R, visualization, torchutils, out_dir, name, n, size, seed, sd, py_seed, steps, lr, max_step):
    """`steps` consecutive iterations of the reference loop (contrast_train.py:128-399): its loop-body text, backward, and its
    own PolyOptimizer built as :90-96 builds it — pins the momentum buffer, the poly LR and the weight update end to end."""
    model = R.Net()
    model.load_state_dict(sd)
    groups = model.get_parameter_groups()                            # before .train(), as contrast_train.py:90
    opt = torchutils.PolyOptimizer([
        {"params": groups[0], "lr": lr, "weight_decay": 5e-4},
        {"params": groups[1], "lr": 2 * lr, "weight_decay": 0},
        {"params": groups[2], "lr": 10 * lr, "weight_decay": 5e-4},
        {"params": groups[3], "lr": 20 * lr, "weight_decay": 0}], lr=lr, weight_decay=5e-4, max_step=max_step)
    model.train()
    masks = []
    for s_ in range(steps):
        masks += [synth.synthetic_dropout_masks(n, (seed + s_) * 2 + 0), synth.synthetic_dropout_masks(n, (seed + s_) * 2 + 1)]
    install_masks(model, masks)
    helpers, body = body_source()
    random.seed(py_seed)
    out = {}
    keys = ["loss", "loss_cls", "loss_er", "loss_ecr", "loss_nce", "loss_intra_nce", "loss_cross_nce", "loss_cross_nce2"]
    for s_ in range(steps):
        img = synth.synthetic_images(n, size, seed + s_)
        lab = synth.synthetic_labels(n, seed + s_)
        ns = {"torch": torch, "F": F, "np": np, "random": random, "visualization": visualization,
              "model": model, "pack": (None, img, lab), "args": types.SimpleNamespace(bg_threshold=0.20)}
        exec(helpers, ns)
        exec(body, ns)
        opt.zero_grad()                                              # contrast_train.py:397-399
        ns["loss"].backward()
        opt.step()
        for k in keys:
            out[f"s{s_}/{k}"] = np.array(float(ns[k].detach()))
        print(name, "step", s_, {k: round(float(ns[k].detach()), 6) for k in keys}, "lr", [g_["lr"] for g_ in opt.param_groups])
    params = dict(model.named_parameters())
    for k in GRAD_KEYS:
        flat = params[k].detach().reshape(-1)
        stepv = max(1, flat.numel() // 4096)
        out["wslice/" + k] = flat[::stepv][:4096].numpy().copy()
        out["wnorm/" + k] = np.array(params[k].detach().double().norm().item())
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), n=n, size=size, seed=seed, py_seed=py_seed, steps=steps, lr=lr,
                        max_step=max_step, lr_final=np.array([g_["lr"] for g_ in opt.param_groups]), **out)
    print("wrote", name)


def eval_golden(out_dir):
    """The reference's evaluator itself (eval.py:13-86, imported from /root/reference) on a small synthetic prediction / ground
    truth set: stores the inputs and its IoU tables for `png` predictions and for `npy` CAM dictionaries at three thresholds."""
    import importlib.util
    import tempfile
    import PIL.Image
    spec = importlib.util.spec_from_file_location("ref_eval", os.path.join(REF, "eval.py"))
    ref_eval = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_eval)
    rng = np.random.default_rng(3)
    names = ["2007_%06d" % (32 + 7 * i) for i in range(6)]
    tmp = tempfile.mkdtemp()
    pd_, gd, nd = (os.path.join(tmp, x) for x in ("pred", "gt", "npy"))
    for d in (pd_, gd, nd):
        os.makedirs(d)
    out = {"names": np.array(names)}
    for i, nm in enumerate(names):
        h, w = 23 + 5 * i, 31 + 3 * i
        cls = rng.choice(np.arange(1, 21), size=1 + i % 3, replace=False)
        gt = np.zeros((h, w), np.uint8)
        for c in cls:
            y0, x0 = rng.integers(0, h // 2), rng.integers(0, w // 2)
            gt[y0:y0 + h // 2, x0:x0 + w // 2] = c
        gt[rng.random((h, w)) < 0.05] = 255                          # ignore label
        pred = gt.copy()
        pred[pred == 255] = 0
        flip = rng.random((h, w)) < 0.2
        pred[flip] = rng.choice(np.concatenate([[0], cls, [int(rng.integers(1, 21))]]), size=int(flip.sum()))
        cams = {int(c) - 1: (rng.random((h, w)).astype(np.float32) * (0.3 + 0.7 * (gt == c))).astype(np.float32) for c in cls}
        PIL.Image.fromarray(gt).save(os.path.join(gd, nm + ".png"))
        PIL.Image.fromarray(pred.astype(np.uint8)).save(os.path.join(pd_, nm + ".png"))
        np.save(os.path.join(nd, nm + ".npy"), cams)
        out[f"gt/{nm}"] = gt
        out[f"pred/{nm}"] = pred.astype(np.uint8)
        out[f"camkeys/{nm}"] = np.array(sorted(cams))
        out[f"cams/{nm}"] = np.stack([cams[k] for k in sorted(cams)])
    cats = ref_eval.categories + ["mIoU"]
    res = ref_eval.do_python_eval(pd_, gd, names, 21, "png", 1.0)
    out["iou/png"] = np.array([res[c] for c in cats])
    for t in (0.1, 0.26, 0.5):
        res = ref_eval.do_python_eval(nd, gd, names, 21, "npy", t)
        out["iou/npy_t%.2f" % t] = np.array([res[c] for c in cats])
    np.savez_compressed(os.path.join(out_dir, "eval_ref.npz"), **out)
    print("wrote eval_ref", {k: float(v[-1]) for k, v in out.items() if k.startswith("iou/")})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="", help="comma list of: fwd,step,step448,step_edge,sgd,topk,infer,infer_ms,infer_full,multistep,eval,refbf16,margins")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.manual_seed(0)
    R, visualization, torchutils = load_reference()
    sd = synth.procedural_state_dict(0)
    todo = a.only.split(",") if a.only else ["fwd", "step", "sgd", "topk", "infer"]
    if "fwd" in todo:
        fwd_golden(R, a.out, "net_fwd_eval_104x72", 1, (104, 72), 11, sd)
        fwd_golden(R, a.out, "net_fwd_eval_128", 2, 128, 12, sd)
    if "step" in todo:
        step_golden(R, visualization, a.out, "step_S160_N2", 2, 160, 21, sd, py_seed=7)
        step_golden(R, visualization, a.out, "step_S128_N3", 3, 128, 22, sd, py_seed=8)
    if "step448" in todo:                            # BASELINE config 2's real resolution (56 x 56 maps, 448 -> 128 resize), batch of 2
        step_golden(R, visualization, a.out, "step_S448_N2", 2, 448, 23, sd, py_seed=6)
        # (seed 23 turned out to hold a near-tie at the top-32 boundary of one class's prototype — kept as a documented case, tests/test_gpu_loss.py;
        #  a second draw at the same resolution:)
        step_golden(R, visualization, a.out, "step_S448_N2_b", 2, 448, 24, sd, py_seed=6)
    if "step_edge" in todo or "step" in todo:
        # an image with all twenty classes carries large gradients through dozens of ReLU pre-activations that sit within f32
        # summation noise of zero (scripts/relu_near_ties.py): backbone gradients of this fixture are compared at 5e-2, the
        # scalars and head gradients at the usual bars (tests/test_gpu_loss.py)
        step_golden(R, visualization, a.out, "step_edge_S64_N3", 3, 64, 16, sd, py_seed=5, edge_labels=True)
    if "sgd" in todo:
        sgd_golden(torchutils, a.out)
    if "topk" in todo:
        topk_pattern_golden(a.out)
    if "infer" in todo:
        infer_golden(R, a.out, sd)
    if "infer_ms" in todo:                           # odd, non-trivial sizes + BASELINE config 5's real geometry (375 x 500: inputs up to 750 x 1000)
        infer_golden(R, a.out, sd, "infer_125x94", 125, 94, (0, 7, 14), seed0=50)
        infer_golden(R, a.out, sd, "infer_188x250", 188, 250, (5,), seed0=60)
    if "infer_full" in todo:
        infer_golden(R, a.out, sd, "infer_375x500", 375, 500, (1, 16), seed0=70, store_stride=3)
    if "refbf16" in todo:                            # the reference itself in bfloat16: the envelope of the benchmarked mode
        step_refbf16_golden(R, visualization, a.out, "step_S160_N2", 2, 160, 21, sd, py_seed=7)
        step_refbf16_golden(R, visualization, a.out, "step_S448_N2", 2, 448, 23, sd, py_seed=6)
        step_refbf16_golden(R, visualization, a.out, "step_S448_N2_b", 2, 448, 24, sd, py_seed=6)
        infer_golden(R, a.out, sd, "infer_188x250", 188, 250, (5,), seed0=60, extras="refbf16")
    if "margins" in todo:                            # the reference's own near-tie margins (prototype top-32 boundary, pseudo-labels, arg-max maps)
        step_margins_golden(R, visualization, a.out, "step_S448_N2", 2, 448, 23, sd, py_seed=6)
        step_margins_golden(R, visualization, a.out, "step_S448_N2_b", 2, 448, 24, sd, py_seed=6)
        infer_golden(R, a.out, sd, "infer_125x94", 125, 94, (0, 7, 14), seed0=50, extras="margins")
        infer_golden(R, a.out, sd, "infer_188x250", 188, 250, (5,), seed0=60, extras="margins")
        infer_golden(R, a.out, sd, "infer_375x500", 375, 500, (1, 16), seed0=70, extras="margins")
    if "multistep" in todo:
        multistep_golden(R, visualization, torchutils, a.out, "step_S128_N3_x3", 3, 128, 51, sd, py_seed=9, steps=3,
                         lr=float(os.environ.get("WSEG_GOLDEN_LR", "3e-6")), max_step=10)
    if "eval" in todo:
        eval_golden(a.out)


if __name__ == "__main__":
    main()
