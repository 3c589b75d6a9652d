"""The module contract the reference's drivers rely on beyond `forward`:

* contrast_infer.py:69-73 — `ThreadPool(num_workers=8)` workers call ONE module concurrently under `no_grad`
  (tool/pyutils.py:76-120); with one GPU all eight tasks hit the same instance, the first call included;
* contrast_infer.py:47 — `nn.parallel.replicate(model, devices)` and a call on the replica;
* contrast_train.py:108 — `nn.DataParallel(model).cuda()` and a call through the wrapper.

All of them must give the serial result of `model(x)` bit for bit."""
from multiprocessing.pool import ThreadPool

import pytest
import torch

pytestmark = pytest.mark.gpu


def _fresh(proc_sd, prec):
    from wseg_amd.resnet38_contrast import Net
    m = Net(precision=prec)
    m.load_state_dict(proc_sd)
    m.eval()
    m.cuda()
    return m


def _inputs():
    from wseg_amd import synth
    sizes = [(40, 56), (40, 56), (64, 48), (64, 48), (72, 104), (72, 104), (33, 47), (96, 96)]    # odd and even, pairs as the MSF stack has
    return [synth.synthetic_images(1, s, 60 + i).cuda() for i, s in enumerate(sizes)]


def _same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_eight_threads_on_one_fresh_module(proc_sd, prec):
    """The FIRST forward of a fresh module builds the flat weight buffer and the packs: eight threads entering it together
    must serialise on that (Engine.lock) and every thread must get the serial answer."""
    xs = _inputs()
    ref_model = _fresh(proc_sd, prec)
    with torch.no_grad():
        ref = [tuple(t.clone() for t in ref_model(x)) for x in xs]
    for attempt in range(2):                                 # a fresh module each time: the race is on the first call
        m = _fresh(proc_sd, prec)

        def work(i):
            with torch.no_grad():
                out = m(xs[i])
                torch.cuda.current_stream().synchronize()
                return tuple(t.clone() for t in out)
        with ThreadPool(processes=8) as pool:
            got = pool.map(work, range(8))
        for i in range(8):
            assert _same(got[i], ref[i]), (attempt, i)
        with ThreadPool(processes=8) as pool:                # and again on the warm module, repeatedly
            for _ in range(3):
                got = pool.map(work, range(8))
                for i in range(8):
                    assert _same(got[i], ref[i]), ("warm", i)


def test_replicate_and_data_parallel(proc_sd):
    """A replica is a shallow copy of the module whose parameters are plain tensors (aliases of the original's on its
    device): it gets an engine of its own (Net._engine) and must compute from ITS tensors — both before and after the
    original has built its flat buffer — and `nn.DataParallel(model)(x)` must equal `model(x)`."""
    x = _inputs()[4]
    m = _fresh(proc_sd, "fp32")
    rep0 = torch.nn.parallel.replicate(m, [0], detach=True)[0]           # original not flattened yet: replica builds its own buffers
    assert rep0._engine is not m._engine and rep0._engine.net is rep0
    with torch.no_grad():
        a = rep0(x)
        ref = m(x)                                                       # (flattens the original)
        rep1 = torch.nn.parallel.replicate(m, [0])[0]                    # parameters now alias the original's flat buffer
        b = rep1(x)
        assert rep1._engine.delegate is m._engine
        c = torch.nn.DataParallel(m)(x)
    assert _same(a, ref) and _same(b, ref) and _same(c, ref)
    # on its own device a replica's parameters are ALIASES of the original's storage (torch hands device 0 the source tensors), so
    # with stock modules an in-place update of the original shows through the replica: the aliasing replica must follow it too
    with torch.no_grad():
        m.fc8.weight.mul_(2.0)
        moved = m(x)
        follows = rep1(x)
    assert not torch.equal(moved[0], ref[0])
    assert _same(follows, moved)


def test_multiscale_inference_through_replica_threads(proc_sd, golden_dir):
    """contrast_infer.py:47-73 as the reference runs it: replicate, then the 8 (scale, flip) forwards from a thread pool on the
    replica; the summed, normalised CAM must equal wseg_amd.infer.infer_image on the plain module bit for bit."""
    import numpy as np
    import os
    from wseg_amd import _lib as L, synth
    from wseg_amd.infer import infer_image
    g = np.load(os.path.join(golden_dir, "infer_1img.npz"))
    H, W = int(g["H"]), int(g["W"])
    lab = torch.from_numpy(g["label"])
    imgs = []
    for si, s in enumerate([0.5, 1.0, 1.5, 2.0]):
        im = synth.synthetic_images(1, (int(np.round(H * s)), int(np.round(W * s))), 40 + si)
        imgs += [im, torch.flip(im, dims=[3])]
    m = _fresh(proc_sd, "fp32")
    norm_ref, pred_ref, _ = infer_image(m, imgs, lab, (H, W))
    rep = torch.nn.parallel.replicate(m, [0])[0]

    def work(i):
        with torch.no_grad():
            out = rep(imgs[i].cuda())[1][0, 1:].contiguous()
            torch.cuda.current_stream().synchronize()
            return out
    with ThreadPool(processes=8) as pool:
        maps = pool.map(work, range(8))
    labd = lab.cuda().float().contiguous()
    sum_cam = torch.zeros(20, H, W, device="cuda")
    for i, mp in enumerate(maps):
        L.resize_planar_fwd(mp, sum_cam, 20, mp.shape[1], mp.shape[2], H, W, False, plane_mul=labd, flip_x=(i % 2 == 1), accumulate=True)
    stats = torch.empty(20, 6, device="cuda")
    L.plane_stats(sum_cam, stats, 20, H * W)
    norm = torch.empty(20, H, W, device="cuda")
    pred = torch.empty(H, W, device="cuda", dtype=torch.uint8)
    L.infer_finish(sum_cam, stats, 0.26, norm, pred, H * W)
    assert torch.equal(pred, pred_ref)
    assert float((norm - norm_ref).abs().max()) <= 1e-5             # (infer_image batches the pairs: same maps up to tile choice)
    assert int((pred.cpu().numpy() != g["pred"]).sum()) == 0            # and the reference's own fixture
