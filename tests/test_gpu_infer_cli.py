"""GPU: multi-scale inference post-process against the reference fixture, the 2-rank prototype path on one GPU,
and CLI plumbing (BASELINE config 1: a few synthetic VOC-format JPEGs, batch_size 2)."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_infer_matches_reference_fixture(golden_dir, proc_sd):
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
    m = Net(precision="fp32"); m.load_state_dict(proc_sd); m.cuda(); m.eval()
    norm_cam, pred, cam_dict = infer_image(m, imgs, lab, (H, W), 0.26)
    np.testing.assert_allclose(norm_cam.cpu().numpy(), g["norm_cam"], rtol=2e-4, atol=2e-5)
    mism = float((pred.cpu().numpy() != g["pred"]).mean())
    assert mism == 0.0, mism                                # CAM argmax: bit-exact goal
    assert sorted(cam_dict.keys()) == [3, 11]


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


def test_cli_plumbing_train_then_infer(tmp_path, monkeypatch):
    """BASELINE.json config 1 on the GPU path: 4 synthetic VOC-format JPEGs, batch_size 2, 1 epoch -> contrast.pth ->
    contrast_infer writes <name>.npy / <name>.png in the reference's formats."""
    import PIL.Image
    from wseg_amd import contrast_infer, contrast_train, synth
    root = tmp_path / "VOC2012"; (root / "JPEGImages").mkdir(parents=True)
    names = [f"2007_00000{i}" for i in range(4)]
    rng = np.random.default_rng(0)
    for n in names:
        PIL.Image.fromarray(rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)).save(root / "JPEGImages" / (n + ".jpg"))
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(f"/JPEGImages/{n}.jpg /SegmentationClassAug/{n}.png" for n in names) + "\n")
    np.save(tmp_path / "cls_labels.npy", {n: synth.synthetic_labels(4, 0)[i].numpy() for i, n in enumerate(names)}, allow_pickle=True)
    monkeypatch.chdir(tmp_path)
    contrast_train.main(["--weights", "procedural", "--batch_size", "2", "--max_epoches", "1", "--train_list", str(lst),
                         "--voc12_root", str(root), "--labels", str(tmp_path / "cls_labels.npy"), "--crop_size", "128",
                         "--num_workers", "0", "--session_name", "t", "--lr", "1e-5", "--precision", "bf16"])
    ckpt = tmp_path / "result" / "t" / "contrast.pth"
    assert ckpt.exists()
    sd = torch.load(ckpt, weights_only=True)
    assert len(sd) == 233 and tuple(sd["fc8.weight"].shape) == (21, 4096, 1, 1)
    contrast_infer.main(["--weights", str(ckpt), "--infer_list", str(lst), "--voc12_root", str(root), "--labels",
                         str(tmp_path / "cls_labels.npy"), "--out_cam", str(tmp_path / "cam"), "--out_cam_pred", str(tmp_path / "pred"),
                         "--num_workers", "0", "--precision", "bf16"])
    d = np.load(tmp_path / "cam" / (names[0] + ".npy"), allow_pickle=True).item()
    assert all(v.shape == (96, 128) and v.dtype == np.float32 for v in d.values()) and len(d) >= 1
    png = np.asarray(PIL.Image.open(tmp_path / "pred" / (names[0] + ".png")))
    assert png.shape == (96, 128) and png.dtype == np.uint8 and png.max() <= 20
