"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/wseg_hip.h declares,
the PolyOptimizer host logic matches the reference fixture, and the VOC-format host plumbing parses the
reference's file formats.  No compute kernel is called."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "wseg_amd", "libwseg_hip.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["bash", os.path.join(ROOT, "wseg_amd", "csrc", "build.sh")])
    return ctypes.CDLL(LIB)


def test_cabi_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "wseg_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(wseg_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 35
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    lib.wseg_version.restype = ctypes.c_int
    assert lib.wseg_version() >= 100
    lib.wseg_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.wseg_last_error(), bytes)


def test_cabi_rejects_bad_descriptor_without_touching_a_gpu(lib):
    from wseg_amd import _lib as L
    d = L.ConvDesc()                      # all-null descriptor: the host-side shape check must refuse it
    assert lib.wseg_conv_igemm(ctypes.byref(d), None) != 0
    assert b"null" in lib.wseg_last_error()
    w = L.WgradDesc()
    assert lib.wseg_conv_wgrad(ctypes.byref(w), None) != 0


def test_poly_optimizer_matches_reference_fixture(golden_dir, lib):
    """wseg_amd.optim.PolyOptimizer on plain (non flat-backed) CPU parameters takes the torch.optim.SGD route:
    the momentum quirk, per-group weight decay, poly LR and `global_step` must match tool/torchutils.py:11-33."""
    from wseg_amd.optim import PolyOptimizer
    g = np.load(os.path.join(golden_dir, "sgd_3steps.npz"))
    ps = [torch.nn.Parameter(torch.from_numpy(g[f"p{i}_init"]).clone()) for i in range(3)]
    opt = PolyOptimizer([{"params": [ps[0]], "lr": 0.01, "weight_decay": 5e-4}, {"params": [ps[1]], "lr": 0.02, "weight_decay": 0},
                         {"params": [ps[2]], "lr": 0.1, "weight_decay": 5e-4}, {"params": [], "lr": 0.2, "weight_decay": 0}],
                        lr=0.01, weight_decay=5e-4, max_step=10)
    assert opt.param_groups[0]["momentum"] == 5e-4
    for s in range(3):
        for i, p in enumerate(ps):
            p.grad = torch.from_numpy(g[f"g{s}_{i}"]).clone()
        if s == 1:
            ps[1].grad = None
        opt.step()
    assert opt.global_step == 3
    for i in range(3):
        np.testing.assert_allclose(ps[i].detach().numpy(), g[f"p{i}_final"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose([gr["lr"] for gr in opt.param_groups], g["lr_final"], rtol=1e-12)


def test_net_module_contract(lib):
    """Constructor, state_dict names/shapes, parameter groups and train() freezing — no forward."""
    import contextlib
    import io
    from wseg_amd import arch
    from wseg_amd.resnet38_contrast import Net
    m = Net(precision="fp32")
    spec = arch.state_dict_spec()
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec.keys()) and len(sd) == 233
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in sd)
    with contextlib.redirect_stdout(io.StringIO()):
        groups = m.get_parameter_groups()
    assert [len(x) for x in groups] == [43, 0, 5, 0]
    m.train()
    assert sum(1 for p in m.parameters() if p.requires_grad) == 40
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 105013824
    assert not any(mod.training for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32))                       # no CPU fallback


def test_voc_host_plumbing(tmp_path, lib):
    import PIL.Image
    from wseg_amd import data as wdata
    from wseg_amd.resnet38_contrast import Normalize
    root = tmp_path / "VOC2012"
    (root / "JPEGImages").mkdir(parents=True)
    names = ["2007_000032", "2007_000039", "2008_000123"]
    rng = np.random.default_rng(0)
    for i, n in enumerate(names):
        PIL.Image.fromarray(rng.integers(0, 256, (60 + 10 * i, 90, 3), dtype=np.uint8)).save(root / "JPEGImages" / (n + ".jpg"))
    lst = tmp_path / "train.txt"
    lst.write_text("\n".join(f"/JPEGImages/{n}.jpg /SegmentationClassAug/{n}.png" for n in names) + "\n")
    labels = {n: np.eye(20, dtype=np.float32)[i] for i, n in enumerate(names)}
    np.save(tmp_path / "cls_labels.npy", labels, allow_pickle=True)
    assert wdata.load_img_name_list(str(lst)) == names
    norm = Normalize()
    model_stub = type("M", (), {"normalize": norm})()
    ds = wdata.VOC12ClsDataset(str(lst), str(root), str(tmp_path / "cls_labels.npy"), wdata.train_transform(model_stub, 64))
    name, img, lab = ds[1]
    assert name == names[1] and tuple(img.shape) == (3, 64, 64) and img.dtype == torch.float32 and lab[1] == 1
    msf = wdata.VOC12ClsDatasetMSF(str(lst), str(root), str(tmp_path / "cls_labels.npy"), scales=[0.5, 1.0, 1.5, 2.0],
                                   inter_transform=[np.asarray, norm, wdata.HWC_to_CHW])
    name, imgs, lab = msf[0]
    assert len(imgs) == 8 and imgs[0].shape == (3, 30, 45) and imgs[6].shape == (3, 120, 180)
    np.testing.assert_array_equal(imgs[3], np.flip(imgs[2], -1))
    x = norm(np.full((2, 2, 3), 255, np.uint8))
    np.testing.assert_allclose(x[0, 0], [(1 - 0.485) / 0.229, (1 - 0.456) / 0.224, (1 - 0.406) / 0.225], rtol=1e-6)


def test_cam_miou_eval(tmp_path, lib):
    """wseg_amd.eval (counterpart of the reference's eval.py): IoU = TP/(T+P-TP), 255 ignored, npy-dict + bg threshold."""
    import PIL.Image
    from wseg_amd import eval as weval
    gt_dir, pr_dir, npy_dir = tmp_path / "gt", tmp_path / "pred", tmp_path / "cam"
    for d in (gt_dir, pr_dir, npy_dir):
        d.mkdir()
    rng = np.random.default_rng(0)
    names = ["2007_000001", "2007_000002"]
    TP = np.zeros(21); P = np.zeros(21); T = np.zeros(21)
    for n in names:
        gt = rng.choice([0, 3, 15, 255], size=(20, 30), p=[0.5, 0.2, 0.2, 0.1]).astype(np.uint8)
        cams = {2: rng.random((20, 30), dtype=np.float32), 14: rng.random((20, 30), dtype=np.float32)}
        np.save(npy_dir / (n + ".npy"), cams, allow_pickle=True)
        pred = np.argmax(np.stack([np.full((20, 30), 0.4, np.float32), np.zeros((20, 30), np.float32), np.zeros((20, 30), np.float32),
                                   cams[2]] + [np.zeros((20, 30), np.float32)] * 11 + [cams[14]] + [np.zeros((20, 30), np.float32)] * 5), 0).astype(np.uint8)
        PIL.Image.fromarray(gt).save(gt_dir / (n + ".png")); PIL.Image.fromarray(pred).save(pr_dir / (n + ".png"))
        for c in range(21):                                      # the reference's per-class loop (eval.py:44-52)
            cal = gt < 255
            P[c] += np.sum((pred == c) * cal); T[c] += np.sum((gt == c) * cal); TP[c] += np.sum((gt == c) * (pred == gt) * cal)
    ref_miou = np.mean(TP / (T + P - TP + 1e-10)) * 100
    a = weval.do_eval(names, str(pr_dir), str(gt_dir), 'png')
    b = weval.do_eval(names, str(npy_dir), str(gt_dir), 'npy', 0.4)
    assert abs(a['mIoU'] - ref_miou) < 1e-9 and abs(b['mIoU'] - ref_miou) < 1e-9
    assert a['bird'] > 0 and a['person'] > 0 and a['aeroplane'] == 0


def test_gradient_buckets_tile_the_flat_buffer():
    """The overlap buckets of the gradient all-reduce (train.py) are contiguous, cover flat_g exactly once, and each
    holds only parameters whose gradients are complete when its trigger block has been processed."""
    import contextlib
    import io
    import torch
    from wseg_amd import arch
    from wseg_amd.resnet38_contrast import Net
    net = Net()
    with contextlib.redirect_stdout(io.StringIO()):
        net.get_parameter_groups()
    eng = net._engine
    eng.ensure_flat(torch.device("cpu"))
    buckets = eng.grad_buckets()
    order = [b[0] for b in arch.BLOCKS if b[0] not in arch.FROZEN_BLOCKS]          # forward order; backward runs it reversed
    spans = sorted(buckets.values())
    assert spans[0][0] == 0 and spans[-1][1] == eng.flat_g.numel()
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for trigger, (lo, hi) in buckets.items():
        done = set(order[order.index(trigger):])                                   # blocks processed when `trigger` completes
        for name, (off, n) in eng.offsets.items():
            if lo <= off < hi:
                blk = name.split(".")[0]
                assert blk in done or blk in ("fc_proj", "fc8", "f8_3", "f8_4", "f9"), (trigger, name)


def test_eval_matches_the_reference_evaluator(golden_dir, tmp_path, lib):
    """wseg_amd.eval.do_eval against IoU tables produced by the reference's own eval.py:13-86 (`do_python_eval`, run by
    oracle/make_goldens.py on the inputs stored in the fixture): png predictions and npy CAM dictionaries at three background
    thresholds, 255 = ignore, classes missing from an image's dictionary; and both list formats eval.py / voc12 use."""
    import PIL.Image
    from wseg_amd import eval as weval
    g = np.load(os.path.join(golden_dir, "eval_ref.npz"))
    names = [str(n) for n in g["names"]]
    for d in ("pred", "gt", "npy"):
        (tmp_path / d).mkdir()
    for n in names:
        PIL.Image.fromarray(g[f"gt/{n}"]).save(tmp_path / "gt" / (n + ".png"))
        PIL.Image.fromarray(g[f"pred/{n}"]).save(tmp_path / "pred" / (n + ".png"))
        np.save(tmp_path / "npy" / (n + ".npy"), {int(k): v for k, v in zip(g[f"camkeys/{n}"], g[f"cams/{n}"])}, allow_pickle=True)
    cats = weval.CATEGORIES + ["mIoU"]
    res = weval.do_eval(names, str(tmp_path / "pred"), str(tmp_path / "gt"), "png")
    np.testing.assert_allclose([res[c] for c in cats], g["iou/png"], rtol=0, atol=1e-9)
    for t in (0.1, 0.26, 0.5):
        res = weval.do_eval(names, str(tmp_path / "npy"), str(tmp_path / "gt"), "npy", t)
        np.testing.assert_allclose([res[c] for c in cats], g["iou/npy_t%.2f" % t], rtol=0, atol=1e-9)
    bare, paths = tmp_path / "bare.txt", tmp_path / "paths.txt"
    bare.write_text("\n".join(names) + "\n")                                       # ImageSets/Segmentation/train.txt (eval.py:112)
    paths.write_text("\n".join(f"/JPEGImages/{n}.jpg /SegmentationClassAug/{n}.png" for n in names) + "\n")     # voc12/*.txt
    for lst in (bare, paths):
        r = weval.main(["--list", str(lst), "--predict_dir", str(tmp_path / "npy"), "--gt_dir", str(tmp_path / "gt"), "--type", "npy", "--t", "0.26"])
        assert abs(r["mIoU"] - float(g["iou/npy_t0.26"][-1])) < 1e-9


def test_pickled_npy_loader_refuses_foreign_globals(tmp_path):
    """cls_labels.npy / <name>.npy are pickles (voc12/data.py:40-44, contrast_infer.py:82-90): the loader admits numeric numpy
    containers only — a file that names any other global is refused before anything from it runs."""
    import pickle
    from wseg_amd import data as wdata
    from wseg_amd.safe_npy import load_pickled_npy
    good = {"2007_000032": np.arange(20, dtype=np.float32), "2007_000039": np.ones(20, np.float32)}
    np.save(tmp_path / "good.npy", good, allow_pickle=True)
    back = load_pickled_npy(str(tmp_path / "good.npy"))
    assert sorted(back) == sorted(good) and all(np.array_equal(back[k], good[k]) for k in good)
    assert [v.tolist() for v in wdata.load_labels(str(tmp_path / "good.npy"), ["2007_000039"])] == [[1.0] * 20]

    class Evil:
        def __reduce__(self):
            return (os.system, ("touch " + str(tmp_path / "pwned"),))
    np.save(tmp_path / "evil.npy", {"x": Evil()}, allow_pickle=True)
    with pytest.raises(pickle.UnpicklingError):
        load_pickled_npy(str(tmp_path / "evil.npy"))
    with pytest.raises(pickle.UnpicklingError):
        wdata.load_labels(str(tmp_path / "evil.npy"), ["x"])
    assert not (tmp_path / "pwned").exists()
    np.save(tmp_path / "plain.npy", np.arange(6).reshape(2, 3))
    assert load_pickled_npy(str(tmp_path / "plain.npy")).shape == (2, 3)


def test_config1_host_pipeline_at_448(tmp_path, lib):
    """BASELINE config 1's host side at its stated size, without a GPU: 4 synthetic 448 x 448 VOC-format JPEGs -> the training
    transforms (contrast_train.py:64-75) -> DataLoader batches of 2 -> [2,3,448,448] float32 + [2,20] labels; the reference's
    step count (len // batch_size * max_epoches, :88); and the training CLI refuses to run without the HIP device (no CPU path)."""
    import PIL.Image
    from wseg_amd import contrast_train, data as wdata, synth
    from wseg_amd.resnet38_contrast import Net
    root = tmp_path / "VOC2012"
    (root / "JPEGImages").mkdir(parents=True)
    names = [f"2007_00000{i}" for i in range(4)]
    rng = np.random.default_rng(0)
    for n in names:
        PIL.Image.fromarray(rng.integers(0, 256, (448, 448, 3), dtype=np.uint8)).save(root / "JPEGImages" / (n + ".jpg"))
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(f"/JPEGImages/{n}.jpg /SegmentationClassAug/{n}.png" for n in names) + "\n")
    np.save(tmp_path / "cls_labels.npy", {n: synth.synthetic_labels(4, 0)[i].numpy() for i, n in enumerate(names)}, allow_pickle=True)
    model = Net(precision="fp32")
    ds = wdata.VOC12ClsDataset(str(lst), str(root), str(tmp_path / "cls_labels.npy"), wdata.train_transform(model, 448))
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0, drop_last=True)
    batches = list(loader)
    assert len(batches) == 2
    for nm, img, lab in batches:
        assert tuple(img.shape) == (2, 3, 448, 448) and img.dtype == torch.float32 and tuple(lab.shape) == (2, 20)
        assert torch.isfinite(img).all() and float(img.abs().max()) < 3.0
    if not torch.cuda.is_available():
        with pytest.raises((RuntimeError, AssertionError, SystemExit)):
            contrast_train.main(["--weights", "procedural", "--batch_size", "2", "--max_epoches", "1", "--train_list", str(lst),
                                 "--voc12_root", str(root), "--labels", str(tmp_path / "cls_labels.npy"), "--num_workers", "0",
                                 "--session_name", str(tmp_path / "s")])


def test_bench_gpus_flag_starts_that_many_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher must start N ranks as a CHILD process (torch.distributed.run) and hand its
    return code back; with RANK set it must refuse a world size that differs from --gpus (no one-GPU number labelled N GPUs)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []

    class R:
        returncode = 7
    monkeypatch.setattr(bench.subprocess, "run", lambda cmd, env=None: (calls.append((cmd, env)), R())[1])
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and len(calls) == 1
    cmd = calls[0][0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert calls[0][1]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=1" in str(e.value.code) and len(calls) == 1
