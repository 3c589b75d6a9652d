"""The oracle (oracle/*.py, CPU restatement) against fixtures produced by the reference itself
(oracle/make_goldens.py).  These pin the oracle; everything on the GPU is then checked against
the oracle."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import infer as oinfer
from oracle import loss as oloss
from oracle import net as onet
from oracle import optim as ooptim
from wseg_amd import synth


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name", ["net_fwd_eval_104x72", "net_fwd_eval_128"])
def test_net_forward_eval(golden_dir, proc_sd, name):
    g = _load(golden_dir, name)
    size = tuple(int(v) for v in np.atleast_1d(g["size"]))
    size = size[0] if len(size) == 1 else size
    x = synth.synthetic_images(int(g["n"]), size, int(g["seed"]))
    with torch.no_grad():
        cam, cam_rv, f_proj, cam_rv_down = onet.net_forward(x, proc_sd, None)
    # same torch ops in the same order on the same host -> expect (near) bit equality
    np.testing.assert_allclose(cam[..., ::4, ::4].numpy(), g["cam_s"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cam_rv[..., ::4, ::4].numpy(), g["cam_rv_s"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f_proj.numpy(), g["f_proj"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cam_rv_down.numpy(), g["cam_rv_down"], rtol=1e-5, atol=1e-6)
    assert (cam.argmax(1).numpy() == g["cam_argmax"]).mean() > 0.9999
    assert (cam_rv.argmax(1).numpy() == g["cam_rv_argmax"]).mean() > 0.9999


@pytest.mark.parametrize("name", ["step_S160_N2", "step_S128_N3", "step_edge_S64_N3", "step_S448_N2", "step_S448_N2_b"])
def test_train_step_loss_and_grads(golden_dir, proc_sd, name):
    g = _load(golden_dir, name)
    n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
    sd = dict(proc_sd)
    keys = onet.trainable_keys(sd)
    for k in keys:
        sd[k] = sd[k].clone().requires_grad_(True)
    img = synth.synthetic_images(n, size, seed)
    lab = torch.from_numpy(g["label"]) if "label" in g.files else synth.synthetic_labels(n, seed)   # (edge fixture: explicit labels)
    m1 = synth.synthetic_dropout_masks(n, seed * 2 + 0)
    m2 = synth.synthetic_dropout_masks(n, seed * 2 + 1)
    extras = {}
    out = oloss.train_step(img, lab, sd, m1, m2, 0.20, random.Random(py_seed), extras)
    for k in ["loss", "loss_cls", "loss_er", "loss_ecr", "loss_nce", "loss_intra_nce",
              "loss_cross_nce", "loss_cross_nce2"]:
        assert abs(float(out[k]) - float(g["s/" + k])) <= 2e-6 * max(1.0, abs(float(g["s/" + k]))), k
    assert (extras["pseudo1"].numpy() == g["pseudo1"]).all()
    assert (extras["pseudo2"].numpy() == g["pseudo2"]).all()
    np.testing.assert_allclose(extras["protos1"].numpy(), g["protos1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(extras["protos2"].numpy(), g["protos2"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(extras["f1"].detach().numpy()[::7], g["f1_s"], rtol=1e-5, atol=1e-6)
    out["loss"].backward()
    n_with_grad = sum(1 for k in keys if sd[k].grad is not None and sd[k].grad.abs().sum() > 0)
    assert n_with_grad == int(g["n_with_grad"]) == 40
    for key in g.files:
        if not key.startswith("gslice/"):
            continue
        k = key[len("gslice/"):]
        gr = sd[k].grad
        flat = gr.reshape(-1)
        step = max(1, flat.numel() // 4096)
        ref = g[key]
        got = flat[::step][:4096].numpy()
        scale = np.abs(ref).max() + 1e-12
        assert np.abs(got - ref).max() / scale < 1e-4, k
        assert abs(gr.double().norm().item() - float(g["gnorm/" + k])) < 1e-4 * float(g["gnorm/" + k]), k


def test_poly_sgd(golden_dir):
    g = _load(golden_dir, "sgd_3steps")
    assert float(g["momentum"]) == 5e-4          # quirk Q1: weight_decay lands in the momentum slot
    ps = [torch.from_numpy(g[f"p{i}_init"]).clone() for i in range(3)]
    opt = ooptim.PolySGD([
        {"params": [ps[0]], "lr": 0.01, "weight_decay": 5e-4},
        {"params": [ps[1]], "lr": 0.02, "weight_decay": 0},
        {"params": [ps[2]], "lr": 0.1, "weight_decay": 5e-4},
        {"params": [], "lr": 0.2, "weight_decay": 0}], lr=0.01, weight_decay=5e-4, max_step=10)
    for s in range(3):
        grads = {id(p): torch.from_numpy(g[f"g{s}_{i}"]) for i, p in enumerate(ps)}
        if s == 1:
            grads[id(ps[1])] = None
        opt.step(grads)
    for i in range(3):
        np.testing.assert_allclose(ps[i].numpy(), g[f"p{i}_final"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose([gr["lr"] for gr in opt.groups], g["lr_final"], rtol=1e-12)


def test_bg_topk_pattern(golden_dir):
    """Q5: torch.topk on the constant background row returns an implementation-defined index set;
    the fixture records what the reference's torch does, the oracle uses the same torch op."""
    g = _load(golden_dir, "bg_topk_pattern")
    for n in (512, 1024, 4096, 32768):
        i = torch.topk(torch.full((1, n), 0.2), 32, dim=-1)[1][0].numpy()
        assert sorted(i.tolist()) == sorted(g[f"n{n}"].tolist())
        assert sorted(g[f"n{n}"].tolist()) == sorted(g[f"row0_n{n}"].tolist())
    assert sorted(g["n4096"].tolist()) == list(range(32))


def test_infer_postprocess(golden_dir, proc_sd):
    g = _load(golden_dir, "infer_1img")
    H, W = int(g["H"]), int(g["W"])
    lab = torch.from_numpy(g["label"])
    imgs = []
    for si, s in enumerate([0.5, 1.0, 1.5, 2.0]):
        hs, ws = int(np.round(H * s)), int(np.round(W * s))
        im = synth.synthetic_images(1, (hs, ws), 40 + si)
        imgs += [im, torch.flip(im, dims=[3])]
    norm_cam, pred, cam_dict = oinfer.infer_one(imgs, lab, proc_sd, (H, W), 0.26)
    np.testing.assert_allclose(norm_cam, g["norm_cam"], rtol=1e-5, atol=1e-6)
    assert (pred == g["pred"]).mean() > 0.999
    assert sorted(cam_dict.keys()) == [3, 11]


def test_infer_multiscale_odd_size(golden_dir, proc_sd):
    """The restatement against the reference's multi-scale fixture at an odd size (125 x 94: feature maps 16 x 12 ... 32 x 24,
    scales whose rounded sizes are odd); the larger fixtures (188 x 250, 375 x 500) are checked on the GPU only — minutes on CPU."""
    g = _load(golden_dir, "infer_125x94")
    H, W, seed0 = int(g["H"]), int(g["W"]), int(g["seed0"])
    lab = torch.from_numpy(g["label"])
    imgs = []
    for si, s in enumerate([0.5, 1.0, 1.5, 2.0]):
        im = synth.synthetic_images(1, (int(np.round(H * s)), int(np.round(W * s))), seed0 + si)
        imgs += [im, torch.flip(im, dims=[3])]
    norm_cam, pred, cam_dict = oinfer.infer_one(imgs, lab, proc_sd, (H, W), 0.26)
    classes = [int(c) for c in g["classes"]]
    assert sorted(cam_dict.keys()) == classes
    st = int(g["store_stride"])
    np.testing.assert_allclose(norm_cam[classes][:, ::st, ::st], g["norm_cam_present"], rtol=1e-5, atol=1e-6)
    assert (pred == g["pred"]).mean() > 0.9999
    absent = [c for c in range(20) if c not in classes]
    assert np.abs(norm_cam[absent] + 1.0).max() < 1e-6         # a gated class: (0 - 0 - 1e-5) / (0 - 0 + 1e-5)


def test_three_consecutive_steps(golden_dir, proc_sd):
    """oracle train_step + oracle PolySGD over three iterations against the fixture the reference's own loop produced
    (contrast_train.py:128-399 with tool/torchutils.py:11-33): scalars of every step, weights after step 3."""
    g = _load(golden_dir, "step_S128_N3_x3")
    n, size, seed, py_seed, steps, lr = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"]), int(g["steps"]), float(g["lr"])
    sd = {k: v.clone() for k, v in proc_sd.items()}
    keys = onet.trainable_keys(sd)
    scratch = ("f8_3.", "f8_4.", "f9.", "fc8.", "fc_proj.")
    # contrast_train.py:90-96 with Net.get_parameter_groups called BEFORE train(): group 0 holds every conv weight outside the
    # from-scratch layers (the frozen ones included — they never get a gradient), group 2 the from-scratch ones
    conv_keys = [k for k, v in sd.items() if v.dim() == 4]
    g0 = [sd[k] for k in conv_keys if not k.startswith(scratch)]
    g2 = [sd[k] for k in conv_keys if k.startswith(scratch)]
    opt = ooptim.PolySGD([{"params": g0, "lr": lr, "weight_decay": 5e-4}, {"params": [], "lr": 2 * lr, "weight_decay": 0},
                          {"params": g2, "lr": 10 * lr, "weight_decay": 5e-4}, {"params": [], "lr": 20 * lr, "weight_decay": 0}],
                         lr=lr, weight_decay=5e-4, max_step=int(g["max_step"]))
    rng = random.Random(py_seed)
    for s_ in range(steps):
        for k in keys:
            sd[k].requires_grad_(True)
            sd[k].grad = None
        out = oloss.train_step(synth.synthetic_images(n, size, seed + s_), synth.synthetic_labels(n, seed + s_), sd,
                               synth.synthetic_dropout_masks(n, (seed + s_) * 2), synth.synthetic_dropout_masks(n, (seed + s_) * 2 + 1), 0.20, rng)
        out["loss"].backward()
        for k in ["loss", "loss_cls", "loss_er", "loss_ecr", "loss_nce", "loss_intra_nce", "loss_cross_nce", "loss_cross_nce2"]:
            ref = float(g[f"s{s_}/{k}"])
            assert abs(float(out[k]) - ref) <= 5e-6 * max(1.0, abs(ref)), (s_, k, float(out[k]), ref)
        grads = {id(sd[k]): sd[k].grad for k in keys}
        for k in keys:
            sd[k].requires_grad_(False)
        opt.step(grads)
    np.testing.assert_allclose([gr["lr"] for gr in opt.groups], g["lr_final"], rtol=1e-12)
    for key in g.files:
        if key.startswith("wslice/"):
            k = key[7:]
            flat = sd[k].reshape(-1)
            stepv = max(1, flat.numel() // 4096)
            w0 = proc_sd[k].reshape(-1)[::stepv][:4096].double().numpy()
            d_ref, d_got = g[key].astype(np.float64) - w0, flat[::stepv][:4096].double().numpy() - w0
            assert np.abs(d_got - d_ref).max() <= 1e-3 * np.abs(d_ref).max(), k
