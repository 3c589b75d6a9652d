"""GPU: multi-scale inference post-process against the reference fixture, the 2-rank prototype path on one GPU,
and CLI plumbing (BASELINE config 1: a few synthetic VOC-format JPEGs, batch_size 2)."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_infer_matches_reference_fixture(golden_dir, proc_sd, prec):
    from wseg_amd import synth
    from wseg_amd.infer import infer_image
    from wseg_amd.resnet38_contrast import Net
    g = np.load(os.path.join(golden_dir, "infer_1img.npz"))
    H, W = int(g["H"]), int(g["W"])
    lab = torch.from_numpy(g["label"])
    imgs = []
    for si, s in enumerate([0.5, 1.0, 1.5, 2.0]):
        im = synth.synthetic_images(1, (int(np.round(H * s)), int(np.round(W * s))), 40 + si)
        imgs += [im, torch.flip(im, dims=[3])]
    m = Net(precision=prec); m.load_state_dict(proc_sd); m.cuda(); m.eval()
    norm_cam, pred, cam_dict = infer_image(m, imgs, lab, (H, W), 0.26)
    np.testing.assert_allclose(norm_cam.cpu().numpy(), g["norm_cam"], rtol=2e-4, atol=2e-5)
    mism = float((pred.cpu().numpy() != g["pred"]).mean())
    assert mism == 0.0, mism                                # CAM argmax: bit-exact goal
    assert sorted(cam_dict.keys()) == [3, 11]


MS_FIXTURES = ["infer_125x94", "infer_188x250", "infer_375x500"]     # odd sizes + BASELINE config 5's real geometry (inputs up to 750 x 1000)
# fp32 (parity mode): exact agreement is the goal and is what the 40 x 56 fixture shows; at these sizes a pixel may differ when the
# reference's own winner / runner-up margin is inside f32 summation noise — the CAM gate of resnet38_contrast.py:46-48 zeroes every
# entry below the per-pixel maximum, a discontinuous function, so two exact-f32 implementations that sum in a different order can
# resolve a near-tie differently.  Such pixels are counted, bounded, and each one is CHECKED to be a near-tie.
#   measured: fp32 1 / 0 / 17 px (8.5e-5, 0, 9.1e-5); bf16x3 (split-bf16 products, ~1e-5 forward deviation) 375x500: 307 px (1.64e-3)
PARITY_MAX_MISMATCH_FRACTION = {"fp32": 1e-4, "bf16x3": 3.3e-3}
PARITY_NEAR_TIE_MARGIN = {"fp32": 2e-3, "bf16x3": 2e-2}
# the REFERENCE's own winner / runner-up margin at a differing pixel: fp32 inside f32 summation noise (measured 1.8e-6 / 0 / 4.3e-5 on the three fixtures); bf16x3 (1e-5
# forward deviation before the discontinuous CAM gate) measured 1.8e-6 / 0 / 1.5e-3 (profiles/r03_parity_measured.txt)
REF_NEAR_TIE_MARGIN = {"fp32": 1e-4, "bf16x3": 3e-3}
# bf16 (throughput mode): mismatching-pixel fraction against the reference's fp32 arg-max maps, bars = 2x the measured values (see the
# test's printed line; procedural weights: near-threshold pixels of the alpha = 0.26 background score and of the class boundaries)
#   measured on an MI355X: 125x94 0.01217 (143 px), 188x250 0.00894 (420 px), 375x500 0.04308 (8077 px)
BF16_MISMATCH_BAR = {"infer_125x94": 0.025, "infer_188x250": 0.018, "infer_375x500": 0.087}
BF16_MIOU_BAR = 89.8         # measured 94.907 over the 6 classes that occur in the reference's maps: bar = 100 - 2 x (100 - measured)


def _msf_inputs(H, W, seed0):
    """The 8 inputs VOC12ClsDatasetMSF yields (voc12/data.py:100-121), as oracle/make_goldens.py built them."""
    from wseg_amd import synth
    imgs = []
    for si, s in enumerate([0.5, 1.0, 1.5, 2.0]):
        im = synth.synthetic_images(1, (int(np.round(H * s)), int(np.round(W * s))), seed0 + si)
        imgs += [im, torch.flip(im, dims=[3])]
    return imgs


def _net(proc_sd, prec):
    from wseg_amd.resnet38_contrast import Net
    m = Net(precision=prec); m.load_state_dict(proc_sd); m.cuda(); m.eval()
    return m


def _write_eval_set(tmp_path, preds, gts, cams):
    import PIL.Image
    for d in ("pred", "gt", "cam"):
        (tmp_path / d).mkdir(exist_ok=True)
    for name in preds:
        PIL.Image.fromarray(preds[name]).save(tmp_path / "pred" / (name + ".png"))
        PIL.Image.fromarray(gts[name]).save(tmp_path / "gt" / (name + ".png"))
        np.save(tmp_path / "cam" / (name + ".npy"), cams[name], allow_pickle=True)


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16"])
def test_multiscale_inference_fixtures_and_cam_miou(golden_dir, proc_sd, tmp_path, prec):
    """contrast_infer.py:49-99 + eval.py:13-86 end to end on the HIP path: three multi-scale fixtures produced by the reference
    itself (odd sizes and a full 375 x 500 image: inputs up to 750 x 1000, 11 750 PCM pixels) -> infer_image -> the files
    contrast_infer writes -> wseg_amd.eval with the REFERENCE's arg-max maps as ground truth.
    fp32 (parity mode): 0 arg-max mismatches, every present class at IoU 100.  bf16 (throughput mode): mismatch fraction under
    a stated bar (2x measured) and the CAM mIoU against the reference's maps reported and asserted."""
    from wseg_amd import eval as weval
    from wseg_amd.infer import infer_image
    m = _net(proc_sd, prec)
    preds, gts, cams, present, measured, margins, ref_margins = {}, {}, {}, set([0]), {}, {}, {}
    for name in MS_FIXTURES:
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        H, W, seed0 = int(g["H"]), int(g["W"]), int(g["seed0"])
        lab = torch.from_numpy(g["label"])
        norm_cam, pred, cam_dict = infer_image(m, _msf_inputs(H, W, seed0), lab, (H, W), 0.26)
        pred = pred.cpu().numpy()
        classes = [int(c) for c in g["classes"]]
        assert sorted(cam_dict.keys()) == classes
        st = int(g["store_stride"])
        got = norm_cam.cpu().numpy()
        mism = float((pred != g["pred"]).mean())
        measured[name] = mism
        if prec != "bf16":
            # every differing pixel must be a near-tie of our own map: top-1 and top-2 of [alpha, present classes] within the margin
            stack = np.concatenate([np.full((1, H, W), 0.26, np.float32), got[classes]], axis=0)
            top2 = np.sort(stack, axis=0)[-2:]
            margin = top2[1] - top2[0]
            bad = pred != g["pred"]
            margins[name] = float(margin[bad].max()) if bad.sum() else 0.0
            assert mism <= PARITY_MAX_MISMATCH_FRACTION[prec], (name, mism)
            assert margins[name] <= PARITY_NEAR_TIE_MARGIN[prec], (name, int(bad.sum()), margins[name])
            # ... and a near-tie OF THE REFERENCE: <name>_margins.npz holds the reference's own top-1 / top-2 margin of [alpha, present classes] at every pixel
            # (float16; oracle/make_goldens.py infer_golden(extras="margins")) — a proof about the reference's numbers, not a bound fitted to ours
            mg = np.load(os.path.join(golden_dir, name + "_margins.npz"))
            assert np.array_equal(mg["pred"], g["pred"])
            ref_margins[name] = float(mg["margin"].astype(np.float32)[bad].max()) if bad.sum() else 0.0
            assert ref_margins[name] <= REF_NEAR_TIE_MARGIN[prec], (name, int(bad.sum()), ref_margins[name])
            vt = {"fp32": 5e-4, "bf16x3": 1e-2}[prec]       # (the CAM gate is discontinuous: measured 2e-4 / 3.3e-3 on the 375 x 500 image)
            np.testing.assert_allclose(got[classes][:, ::st, ::st], g["norm_cam_present"], rtol=vt, atol=vt)
            np.testing.assert_allclose(got[classes].astype(np.float64).sum(axis=(1, 2)), g["sums"], rtol=vt)
        absent = [c for c in range(20) if c not in classes]
        assert float(np.abs(got[absent] - (-1.0)).max()) <= 1e-6                # label gating: absent classes are the constant -1
        preds[name], gts[name] = pred, g["pred"]
        cams[name] = {k: v.cpu().numpy() for k, v in cam_dict.items()}
        present |= set(int(c) for c in np.unique(g["pred"]))             # classes that occur in the ground truth (= the reference's maps)
    print(f"{prec}: arg-max mismatch fraction vs the reference's maps: " + ", ".join(f"{k} {v:.6f} ({int(round(v * preds[k].size))} px)" for k, v in measured.items()))
    if prec == "bf16":
        assert all(measured[k] <= BF16_MISMATCH_BAR[k] for k in MS_FIXTURES), measured
    else:
        print(f"{prec}: largest top-1 / top-2 margin among the differing pixels: ours {margins}, the reference's own {ref_margins}")
    _write_eval_set(tmp_path, preds, gts, cams)
    res = weval.do_eval(MS_FIXTURES, str(tmp_path / "pred"), str(tmp_path / "gt"), "png")
    res_npy = weval.do_eval(MS_FIXTURES, str(tmp_path / "cam"), str(tmp_path / "gt"), "npy", 0.26)
    ious = [res[weval.CATEGORIES[c]] for c in sorted(present)]
    miou_present = float(np.mean(ious))
    print(f"{prec}: CAM mIoU vs the reference's maps over the {len(present)} present classes = {miou_present:.3f} "
          f"(21-class mean as eval.py prints it: {res['mIoU']:.3f})")
    for k in res:                                                               # the .npy route (threshold 0.26 = alpha) is the same arg-max
        assert abs(res[k] - res_npy[k]) <= 1e-9, k
    if prec != "bf16":
        floor = 100.0 - 100.0 * 21 * PARITY_MAX_MISMATCH_FRACTION[prec]
        assert all(v >= floor for v in ious), ious                              # = 100 when no near-tie pixel differs
        assert miou_present >= floor, miou_present
    else:
        assert miou_present >= BF16_MIOU_BAR, miou_present


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_config5_geometry_properties(proc_sd, prec):
    """BASELINE config 5 at its real geometry (375 x 500, all 8 inputs) through properties that need no oracle: shapes and
    finiteness, label gating, value range, and flip covariance — the MSF stack of the mirrored image is the same 8 tensors with
    every (original, flipped) pair swapped, so the result must be the mirrored result (checks the un-flip + pair batching +
    two-segment launch sequence; equal up to the order of the 8-term float sum)."""
    from wseg_amd.infer import infer_image
    m = _net(proc_sd, prec)
    H, W = 375, 500
    imgs = _msf_inputs(H, W, 90)
    lab = torch.zeros(20); lab[[2, 9, 19]] = 1
    norm, pred, cams = infer_image(m, imgs, lab, (H, W))
    assert tuple(norm.shape) == (20, H, W) and tuple(pred.shape) == (H, W) and pred.dtype == torch.uint8
    assert torch.isfinite(norm).all()
    assert sorted(cams) == [2, 9, 19] and int(pred.max()) <= 20
    assert set(np.unique(pred.cpu().numpy()).tolist()) <= {0, 3, 10, 20}
    pres = norm[[2, 9, 19]]
    assert float(pres.max()) <= 1.0 + 1e-6 and float(pres.max()) >= 0.99          # max-normalised to (max - min - 1e-5)/(max - min + 1e-5)
    swapped = [imgs[i ^ 1] for i in range(8)]
    norm_f, pred_f, _ = infer_image(m, swapped, lab, (H, W))
    # (not bit-equal: the 8-term sums run in a different order, and the CAM gate is discontinuous — a near-tie resolved differently moves
    #  the PCM-refined map by O(1 / hw); measured 1.6e-4 in fp32, 0.7e-4 in bf16)
    assert float((norm_f - torch.flip(norm, dims=[2])).abs().max()) <= 1e-3
    assert float((pred_f != torch.flip(pred, dims=[1])).float().mean()) <= 2e-4


def test_prototype_exchange_two_ranks_on_one_gpu():
    """csrc/loss.hip proto_candidates + proto_merge with world=2 (candidate lists of two half-batches stacked as the
    all-gather would) == the single-pass result over the whole batch == the torch semantics of tests/test_dist_gloo.py."""
    from tests.test_dist_gloo import local_candidates, merge
    from wseg_amd import _lib as L
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    n, npix, K = 4, 256, 32
    ncam = torch.rand(n, 21, npix, generator=g); ncam[:, 0] = 0.2; ncam[:, 7] = -1.0; ncam[2:, 9] = -1.0
    feat = torch.randn(n * npix, 128, generator=g)
    tie = torch.arange(K, dtype=torch.int32)

    def cands(nc, ft):
        N = nc.shape[0]
        cv = torch.empty(21, K, device=dev); cf = torch.empty(21, K, 128, device=dev); cc = torch.empty(21, device=dev, dtype=torch.int32)
        L.proto_candidates(nc.contiguous().to(dev), ft.contiguous().to(dev), tie.to(dev), cv, cf, cc, N, npix, K)
        return cv, cf, cc

    cv, cf, cc = cands(ncam, feat)
    p1 = torch.empty(21, 128, device=dev)
    L.proto_merge(cv, cf, cc, p1, 1, K)
    a = cands(ncam[:2], feat[:2 * npix]); b = cands(ncam[2:], feat[2 * npix:])
    p2 = torch.empty(21, 128, device=dev)
    L.proto_merge(torch.stack([a[0], b[0]]), torch.stack([a[1], b[1]]), torch.stack([a[2], b[2]]), p2, 2, K)
    rv, rf, rc = local_candidates(ncam, feat, K, tie.long())
    ref = merge(rv[None], rf[None], rc[None])
    np.testing.assert_allclose(p1.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(p2.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    # the product layout: ONE gathered buffer [world][2 views][block], this view's blocks a rank stride apart (loss_hip.py)
    from wseg_amd.loss_hip import _CAND_L
    gathered = torch.zeros(2, 2, _CAND_L, device=dev)
    for w, (v_, f_, c_) in enumerate((a, b)):
        blk = gathered[w, 1]
        blk[:21 * K] = v_.reshape(-1); blk[21 * K:21 * K * 129] = f_.reshape(-1)
        blk[21 * K * 129:21 * K * 129 + 21].view(torch.int32).copy_(c_)
    base = gathered.view(-1)[_CAND_L:]
    p3 = torch.empty(21, 128, device=dev)
    L.proto_merge(base, base[21 * K:], base[21 * K * 129:].view(torch.int32), p3, 2, K, 2 * _CAND_L)
    np.testing.assert_allclose(p3.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


def test_prototype_merge_eight_ranks():
    """proto_merge at BASELINE config 3's world size (8 x 32 candidates per class, ranked by counting): the eight ranks' candidate lists of
    a 16-image batch == one pass over the whole batch; with values tied ACROSS ranks (the lowest global pixel must win) and a class that is
    constant on every rank (the tie table of rank 0)."""
    from tests.test_dist_gloo import local_candidates, merge
    from wseg_amd import _lib as L
    dev = "cuda"
    g = torch.Generator().manual_seed(8)
    n, npix, K, world = 16, 256, 32, 8
    ncam = (torch.rand(n, 21, npix, generator=g) * 64).round() / 64           # many exact ties, also across ranks
    ncam[:, 0] = 0.2; ncam[:, 7] = -1.0; ncam[4:, 9] = -1.0
    feat = torch.randn(n * npix, 128, generator=g)
    tie = torch.arange(K, dtype=torch.int32)
    per = n // world
    cvs, cfs, ccs = [], [], []
    for r in range(world):
        cv = torch.empty(21, K, device=dev); cf = torch.empty(21, K, 128, device=dev); cc = torch.empty(21, device=dev, dtype=torch.int32)
        L.proto_candidates(ncam[r * per:(r + 1) * per].contiguous().to(dev), feat[r * per * npix:(r + 1) * per * npix].contiguous().to(dev),
                           tie.to(dev), cv, cf, cc, per, npix, K)
        cvs.append(cv); cfs.append(cf); ccs.append(cc)
    p8 = torch.empty(21, 128, device=dev)
    L.proto_merge(torch.stack(cvs), torch.stack(cfs), torch.stack(ccs), p8, world, K)
    rv, rf, rc = local_candidates(ncam, feat, K, tie.long())
    ref = merge(rv[None], rf[None], rc[None])
    np.testing.assert_allclose(p8.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("img_hw,crop,prec,dev_aug", [((96, 128), 128, "bf16", False), ((448, 448), 448, "fp32", False), ((375, 500), 448, "bf16x3", True)])
def test_cli_plumbing_train_then_infer(tmp_path, monkeypatch, img_hw, crop, prec, dev_aug):
    """BASELINE.json config 1 on the GPU path: 4 synthetic VOC-format JPEGs, batch_size 2, 1 epoch -> contrast.pth ->
    contrast_infer writes <name>.npy / <name>.png in the reference's formats.  Second case: the config's stated size
    (448 x 448 JPEGs, --crop_size 448, the reference's default) in the parity precision.  Third case: VOC-sized 375 x 500 JPEGs through the
    device-side augmentation (--device_augment) in the split-bf16 mode."""
    import PIL.Image
    from wseg_amd import contrast_infer, contrast_train, synth
    root = tmp_path / "VOC2012"; (root / "JPEGImages").mkdir(parents=True)
    names = [f"2007_00000{i}" for i in range(4)]
    rng = np.random.default_rng(0)
    for n in names:
        PIL.Image.fromarray(rng.integers(0, 256, img_hw + (3,), dtype=np.uint8)).save(root / "JPEGImages" / (n + ".jpg"))
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(f"/JPEGImages/{n}.jpg /SegmentationClassAug/{n}.png" for n in names) + "\n")
    np.save(tmp_path / "cls_labels.npy", {n: synth.synthetic_labels(4, 0)[i].numpy() for i, n in enumerate(names)}, allow_pickle=True)
    monkeypatch.chdir(tmp_path)
    contrast_train.main(["--weights", "procedural", "--batch_size", "2", "--max_epoches", "1", "--train_list", str(lst),
                         "--voc12_root", str(root), "--labels", str(tmp_path / "cls_labels.npy"), "--crop_size", str(crop),
                         "--num_workers", "0", "--session_name", "t", "--lr", "1e-5", "--precision", prec] + (["--device_augment"] if dev_aug else []))
    ckpt = tmp_path / "result" / "t" / "contrast.pth"
    assert ckpt.exists()
    sd = torch.load(ckpt, weights_only=True)
    assert len(sd) == 233 and tuple(sd["fc8.weight"].shape) == (21, 4096, 1, 1)
    contrast_infer.main(["--weights", str(ckpt), "--infer_list", str(lst), "--voc12_root", str(root), "--labels",
                         str(tmp_path / "cls_labels.npy"), "--out_cam", str(tmp_path / "cam"), "--out_cam_pred", str(tmp_path / "pred"),
                         "--num_workers", "0", "--precision", prec])
    d = np.load(tmp_path / "cam" / (names[0] + ".npy"), allow_pickle=True).item()
    assert all(v.shape == img_hw and v.dtype == np.float32 for v in d.values()) and len(d) >= 1
    png = np.asarray(PIL.Image.open(tmp_path / "pred" / (names[0] + ".png")))
    assert png.shape == img_hw and png.dtype == np.uint8 and png.max() <= 20
    # ... and the evaluator reads what the inference CLI wrote (eval.py:109-136): with the written pngs as ground truth, IoU = 100
    from wseg_amd import eval as weval
    res = weval.main(["--list", str(lst), "--predict_dir", str(tmp_path / "cam"), "--gt_dir", str(tmp_path / "pred"), "--type", "npy", "--t", "0.26"])
    seen = set(np.unique(np.stack([np.asarray(PIL.Image.open(tmp_path / "pred" / (n + ".png"))) for n in names])).tolist())
    assert all(abs(res[weval.CATEGORIES[c]] - 100.0) < 1e-7 for c in seen)
