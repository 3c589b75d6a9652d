"""GPU parity: the HIP implicit-GEMM conv (fwd / dgrad / wgrad / stem) through the C ABI against
torch's CPU convolution on the same seeded inputs (the op the reference's CPU path runs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _nhwc(x):      # [N,C,H,W] -> contiguous [N,H,W,C]
    return x.permute(0, 2, 3, 1).contiguous()


CASES = [
    # N, H, W, IC, OC, k, stride, dil
    (2, 20, 20, 64, 128, 3, 1, 1),
    (1, 23, 17, 128, 64, 3, 1, 2),
    (2, 14, 14, 64, 256, 3, 1, 4),
    (2, 21, 19, 64, 128, 3, 2, 1),
    (2, 20, 20, 128, 128, 1, 1, 1),
    (1, 21, 19, 64, 192, 1, 2, 1),
    (1, 9, 9, 256, 24, 1, 1, 1),
    (1, 13, 11, 256, 512, 3, 1, 2),      # 256x256-tile wgrad path (bf16), odd spatial size
    (2, 10, 10, 320, 256, 1, 1, 1),      # 256-tile path with an IC tail
]


@pytest.mark.parametrize("bm", [64, 128])
@pytest.mark.parametrize("dt", ["f32", "bf16", "x3"])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case, dt, bm):
    """dt x3 = WSEG_F32X3: f32 tensors, products as split-bf16 (hi.hi + lo.hi + hi.lo on the bf16 MFMA, f32 accumulate); the
    activations are split in the kernels, the weight packs arrive pre-split (wseg_pack_x3)."""
    from wseg_amd import _lib as L
    N, H, W, IC, OC, k, s, d = case
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    cdt = L.F32X3 if dt == "x3" else None
    pad = d * (k // 2)
    OH = (H + 2 * pad - d * (k - 1) - 1) // s + 1
    OW = (W + 2 * pad - d * (k - 1) - 1) // s + 1
    x = _rand((N, IC, H, W), 1).to(tdt).float()
    w = _rand((OC, IC, k, k), 2, (2.0 / (IC * k * k)) ** 0.5).to(tdt).float()
    dy = _rand((N, OC, OH, OW), 3).to(tdt).float()
    x.requires_grad_(True)
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, s, pad, d)
    y.backward(dy)
    tol = {"f32": dict(rtol=2e-5, atol=2e-5), "x3": dict(rtol=1e-4, atol=1e-4), "bf16": dict(rtol=2e-2, atol=2e-2)}[dt]

    dev = "cuda"
    xg = _nhwc(x.detach()).to(dev, tdt)
    w_krsc = w.detach().permute(0, 2, 3, 1).contiguous()            # [OC][k][k][IC] f32 master
    wm = w_krsc.to(dev)
    wf = torch.empty(OC, k * k, IC, device=dev, dtype=tdt)
    wt = torch.empty(IC, k * k, OC, device=dev, dtype=tdt)
    L.pack_weights(wm, wf, wt, OC, k * k, IC, OC, IC, L.dtype_code(wf))
    if dt == "x3":
        wf32, wt32 = wf, wt
        wf, wt = torch.empty_like(wf32), torch.empty_like(wt32)
        L.pack_x3(wf32, wf)
        if (OC * 4) % 128 == 0:
            L.pack_x3(wt32, wt)
    # forward
    yg = torch.empty(N, OH, OW, OC, device=dev, dtype=tdt)
    L.conv_igemm(xg, wf, yg, N=N, IH=H, IW=W, IC=IC, OH=OH, OW=OW, OC=OC, KH=k, KW=k, stride=s, dil=d, pad=pad, bm_hint=bm, dtype=cdt)
    np.testing.assert_allclose(yg.float().cpu().numpy(), _nhwc(y.detach()).numpy(), **tol)
    # data gradient (needs OC % (128B/es) == 0 as the reduction dim)
    es = 2 if dt == "bf16" else 4
    dyg = _nhwc(dy).to(dev, tdt)
    if (OC * es) % 128 == 0:
        dxg = torch.empty(N, H, W, IC, device=dev, dtype=tdt)
        L.conv_igemm(dyg, wt, dxg, N=N, IH=OH, IW=OW, IC=OC, OH=H, OW=W, OC=IC, KH=k, KW=k, stride=s, dil=d, pad=pad, mode=1, bm_hint=bm, dtype=cdt)
        np.testing.assert_allclose(dxg.float().cpu().numpy(), _nhwc(x.grad).numpy(), **tol)
    # weight gradient (f32, accumulating)
    dwg = torch.zeros(OC, k * k, IC, device=dev, dtype=torch.float32)
    L.conv_wgrad(xg, dyg, dwg, N=N, IH=H, IW=W, IC=IC, OH=OH, OW=OW, OC=OC, KH=k, KW=k, stride=s, dil=d, pad=pad,
                 tile_hint=256 if bm == 128 else 128, dtype=cdt)
    ref = w.grad.permute(0, 2, 3, 1).reshape(OC, k * k, IC).numpy()
    scale = np.abs(ref).max()
    wtol = {"f32": 2e-5, "x3": 1e-4, "bf16": 1e-2}[dt]
    assert np.abs(dwg.cpu().numpy() - ref).max() / scale < wtol
    # split-K path + accumulation on top of existing content
    L.conv_wgrad(xg, dyg, dwg, N=N, IH=H, IW=W, IC=IC, OH=OH, OW=OW, OC=OC, KH=k, KW=k, stride=s, dil=d, pad=pad, split_k=3, dtype=cdt)
    assert np.abs(dwg.cpu().numpy() - 2 * ref).max() / scale < 2 * wtol


@pytest.mark.parametrize("dt,OC,bm", [("f32", 128, 0), ("bf16", 128, 0), ("bf16", 256, 256), ("bf16", 256, 224), ("bf16", 128, 259)])
def test_conv_epilogues(dt, OC, bm):
    """fused BN-ReLU-dropout second output (forward) and masked-scale (+residual) epilogue (backward);
    bm=256: the 256x256 phase-pipelined kernel (300 rows = one full + one partial row tile)."""
    from wseg_amd import _lib as L
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    N, H, W, IC, k = 2, 15, 10, 64, 3
    dev = "cuda"
    x = _rand((N, IC, H, W), 1).to(tdt).float()
    w = _rand((OC, IC, k, k), 2, 0.06).to(tdt).float()
    res = _rand((N, OC, H, W), 4).to(tdt).float()
    pre = _rand((N, OC, H, W), 5).to(tdt).float()
    scale, shift = _rand((OC,), 6) + 1.5, _rand((OC,), 7)
    drop = (torch.rand(N, OC, generator=torch.Generator().manual_seed(8)) > 0.5).float() * 2
    y = F.conv2d(x, w, None, 1, 1, 1)
    raw = y + pre + res
    act = F.relu(raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) * drop.view(N, OC, 1, 1)
    xg = _nhwc(x).to(dev, tdt)
    wf = w.permute(0, 2, 3, 1).reshape(OC, k * k, IC).contiguous().to(dev, tdt)
    out = torch.empty(N, H, W, OC, device=dev, dtype=tdt)
    out2 = torch.empty(N, H, W, OC, device=dev, dtype=tdt)
    L.conv_igemm(xg, wf, out, out2, N=N, IH=H, IW=W, IC=IC, OH=H, OW=W, OC=OC, KH=k, KW=k, pad=1,
                 r_pre=_nhwc(pre).to(dev, tdt), r_post=_nhwc(res).to(dev, tdt),
                 scale=scale.to(dev), shift=shift.to(dev), drop=drop.to(dev), bm_hint=bm)
    tol = dict(rtol=2e-5, atol=2e-5) if dt == "f32" else dict(rtol=2e-2, atol=3e-2)
    np.testing.assert_allclose(out.float().cpu().numpy(), _nhwc(raw).numpy(), **tol)
    # out2 is computed from the unrounded sum in-kernel; compare loosely in bf16
    np.testing.assert_allclose(out2.float().cpu().numpy(), _nhwc(act).numpy(), **tol)
    # epi 1: (acc + pre) * scale * drop * (mask > 0) + post
    mask = _rand((N, OC, H, W), 9)
    exp = (y + pre) * scale.view(1, -1, 1, 1) * drop.view(N, OC, 1, 1) * (mask > 0).float() + res
    L.conv_igemm(xg, wf, out, N=N, IH=H, IW=W, IC=IC, OH=H, OW=W, OC=OC, KH=k, KW=k, pad=1, epi=1,
                 r_pre=_nhwc(pre).to(dev, tdt), r_post=_nhwc(res).to(dev, tdt), mask=_nhwc(mask).to(dev, tdt),
                 scale=scale.to(dev), drop=drop.to(dev), bm_hint=bm)
    np.testing.assert_allclose(out.float().cpu().numpy(), _nhwc(exp).numpy(), **tol)
    # epi 2 with an output row stride and channel offset (writes into a wider buffer)
    wide = torch.zeros(N, H, W, OC + 128, device=dev, dtype=tdt)
    L.conv_igemm(xg, wf, wide[..., 64:], N=N, IH=H, IW=W, IC=IC, OH=H, OW=W, OC=OC, KH=k, KW=k, pad=1, epi=2, ld_out=OC + 128,
                 bm_hint=bm)
    np.testing.assert_allclose(wide[..., 64:64 + OC].float().cpu().numpy(), _nhwc(F.relu(y)).numpy(), **tol)
    assert float(wide[..., :64].abs().max()) == 0 and float(wide[..., 64 + OC:].abs().max()) == 0


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_stem(dt):
    from wseg_amd import _lib as L
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    N, H, W = 2, 37, 70
    x = _rand((N, 3, H, W), 1)
    w = _rand((64, 3, 3, 3), 2, 0.3)
    scale, shift = _rand((64,), 3) + 1.5, _rand((64,), 4)
    y = F.conv2d(x, w, None, 1, 1)
    act = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    dev = "cuda"
    raw_g = torch.empty(N, H, W, 64, device=dev, dtype=tdt)
    act_g = torch.empty(N, H, W, 64, device=dev, dtype=tdt)
    L.stem_conv(x.to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), scale.to(dev), shift.to(dev), raw_g, act_g, N, H, W, L.dtype_code(raw_g))
    tol = dict(rtol=1e-5, atol=1e-5) if dt == "f32" else dict(rtol=1e-2, atol=2e-2)
    np.testing.assert_allclose(raw_g.float().cpu().numpy(), _nhwc(y).numpy(), **tol)
    np.testing.assert_allclose(act_g.float().cpu().numpy(), _nhwc(act).numpy(), **tol)
    # packed-FMA form on the transposed weights [27][64]: the same products in the same order -> bit-identical
    raw_k, act_k = torch.empty_like(raw_g), torch.empty_like(act_g)
    L.stem_conv_kc(x.to(dev), w.permute(2, 3, 1, 0).reshape(27, 64).contiguous().to(dev), scale.to(dev), shift.to(dev), raw_k, act_k, N, H, W,
                   L.dtype_code(raw_g))
    assert torch.equal(raw_k, raw_g) and torch.equal(act_k, act_g)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv_many_row_tiles(dt):
    """521 row tiles (more than the 512 resident workgroups), odd image size, residual epilogue."""
    from wseg_amd import _lib as L
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    N, H, W, IC, OC = 1, 258, 258, 64, 128
    x = _rand((N, IC, H, W), 1).to(tdt).float()
    w = _rand((OC, IC, 3, 3), 2, 0.05).to(tdt).float()
    res = _rand((N, OC, H, W), 3).to(tdt).float()
    y = F.conv2d(x, w, None, 1, 1, 1) + res
    dev = "cuda"
    yg = torch.empty(N, H, W, OC, device=dev, dtype=tdt)
    wf = w.permute(0, 2, 3, 1).reshape(OC, 9, IC).contiguous().to(dev, tdt)
    L.conv_igemm(_nhwc(x).to(dev, tdt), wf, yg, N=N, IH=H, IW=W, IC=IC, OH=H, OW=W, OC=OC, KH=3, KW=3, pad=1,
                 r_post=_nhwc(res).to(dev, tdt))
    tol = dict(rtol=2e-5, atol=2e-5) if dt == "f32" else dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(yg.float().cpu().numpy(), _nhwc(y).numpy(), **tol)


@pytest.mark.parametrize("dt,bm", [("f32", 0), ("bf16", 0), ("bf16", 256), ("bf16", 2560), ("bf16", 224), ("bf16", 2240), ("bf16", 259)])
@pytest.mark.parametrize("geom", [(3, 1, 2), (3, 2, 1), (1, 2, 1)])
def test_conv_two_row_segments(dt, bm, geom):
    """Two views batched in one launch: rows [0,N*OH*OW) use geometry 1, the rest geometry 2 (fwd, dgrad, wgrad)."""
    from wseg_amd import _lib as L
    k, s, d = geom
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    N, IC, OC = 2, (256 if bm >= 224 else 64), 256      # bm>=224: dgrad (OC = IC) takes the big-tile kernels too
    (H1, W1), (H2, W2) = (20, 18), ((8, 12) if bm in (2560, 2240) else (9, 11))   # 2560 / 2240: even sizes in both segments (parity-permuted s2 dgrad)
    bm = {2560: 256, 2240: 224}.get(bm, bm)
    pad = d * (k // 2)
    osz = lambda h: (h + 2 * pad - d * (k - 1) - 1) // s + 1
    xs = [_rand((N, IC, H1, W1), 1).to(tdt).float().requires_grad_(True), _rand((N, IC, H2, W2), 2).to(tdt).float().requires_grad_(True)]
    w = _rand((OC, IC, k, k), 3, (2.0 / (IC * k * k)) ** 0.5).to(tdt).float().requires_grad_(True)
    ys = [F.conv2d(x, w, None, s, pad, d) for x in xs]
    dys = [_rand(tuple(y.shape), 4 + i).to(tdt).float() for i, y in enumerate(ys)]
    sum((y * dy).sum() for y, dy in zip(ys, dys)).backward()
    dev = "cuda"
    rows = lambda t: _nhwc(t).reshape(-1, t.shape[1])
    xj = torch.cat([rows(x.detach()) for x in xs]).to(dev, tdt)
    dyj = torch.cat([rows(dy) for dy in dys]).to(dev, tdt)
    wm = w.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    wf = torch.empty(OC, k * k, IC, device=dev, dtype=tdt); wt = torch.empty(IC, k * k, OC, device=dev, dtype=tdt)
    L.pack_weights(wm, wf, wt, OC, k * k, IC, OC, IC, L.dtype_code(wf))
    O1, O2 = (osz(H1), osz(W1)), (osz(H2), osz(W2))
    seg_f = (H2, W2, O2[0], O2[1])
    yg = torch.empty(dyj.shape[0], OC, device=dev, dtype=tdt)
    L.conv_igemm(xj, wf, yg, N=N, IH=H1, IW=W1, IC=IC, OH=O1[0], OW=O1[1], OC=OC, KH=k, KW=k, stride=s, dil=d, pad=pad, seg2=seg_f, bm_hint=bm)
    tol = dict(rtol=2e-5, atol=2e-5) if dt == "f32" else dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(yg.float().cpu().numpy(), torch.cat([rows(y.detach()) for y in ys]).numpy(), **tol)
    dxg = torch.empty(xj.shape[0], IC, device=dev, dtype=tdt)
    L.conv_igemm(dyj, wt, dxg, N=N, IH=O1[0], IW=O1[1], IC=OC, OH=H1, OW=W1, OC=IC, KH=k, KW=k, stride=s, dil=d, pad=pad, mode=1,
                 seg2=(O2[0], O2[1], H2, W2), bm_hint=bm)
    np.testing.assert_allclose(dxg.float().cpu().numpy(), torch.cat([rows(x.grad) for x in xs]).numpy(), **tol)
    ref = w.grad.permute(0, 2, 3, 1).reshape(OC, k * k, IC).numpy()
    for hint in (128, 256):
        dwg = torch.zeros(OC, k * k, IC, device=dev, dtype=torch.float32)
        L.conv_wgrad(xj, dyj, dwg, N=N, IH=H1, IW=W1, IC=IC, OH=O1[0], OW=O1[1], OC=OC, KH=k, KW=k, stride=s, dil=d, pad=pad,
                     seg2=seg_f, tile_hint=hint, split_k=2)
        assert np.abs(dwg.cpu().numpy() - ref).max() / np.abs(ref).max() < (2e-5 if dt == "f32" else 1e-2)


CASES256 = [
    # N, H, W, IC, OC, k, stride, dil      (bf16, OC % 256 == 0: the 256x256 phase-pipelined kernel)
    (2, 20, 19, 128, 512, 3, 1, 2),      # 760 rows: 2 full + 1 partial row tile, 2 column tiles, 18 K-tiles
    (1, 33, 31, 256, 256, 3, 2, 1),      # stride 2 (dgrad with inexact divisions), dgrad through the same kernel
    (2, 16, 16, 192, 256, 1, 1, 1),      # 1x1, 3 K-tiles per tap
    (2, 17, 15, 64, 256, 1, 2, 1),       # 1x1 stride 2, ONE K-tile (prologue-only pipeline)
    (1, 12, 12, 64, 256, 3, 1, 4),       # dilation 4 on a small map: most taps are padding
    (3, 9, 9, 128, 768, 3, 1, 1),        # 243 rows (< one tile), 3 column tiles
    (2, 32, 28, 256, 256, 3, 2, 1),      # stride 2, even sizes: dgrad walks the rows in parity-class order (pure + mixed tiles)
    (2, 24, 20, 256, 256, 1, 2, 1),      # 1x1 stride 2: three of the four parity classes have no tap at all
]


@pytest.mark.parametrize("case", CASES256)
def test_conv256_fwd_dgrad(case):
    from wseg_amd import _lib as L
    N, H, W, IC, OC, k, s, d = case
    tdt = torch.bfloat16
    pad = d * (k // 2)
    OH = (H + 2 * pad - d * (k - 1) - 1) // s + 1
    OW = (W + 2 * pad - d * (k - 1) - 1) // s + 1
    x = _rand((N, IC, H, W), 1).to(tdt).float().requires_grad_(True)
    w = _rand((OC, IC, k, k), 2, (2.0 / (IC * k * k)) ** 0.5).to(tdt).float().requires_grad_(True)
    dy = _rand((N, OC, OH, OW), 3).to(tdt).float()
    y = F.conv2d(x, w, None, s, pad, d)
    y.backward(dy)
    dev = "cuda"
    xg = _nhwc(x.detach()).to(dev, tdt)
    wm = w.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    wf = torch.empty(OC, k * k, IC, device=dev, dtype=tdt)
    wt = torch.empty(IC, k * k, OC, device=dev, dtype=tdt)
    L.pack_weights(wm, wf, wt, OC, k * k, IC, OC, IC, L.dtype_code(wf))
    tol = dict(rtol=2e-2, atol=2e-2)
    yg = torch.empty(N, OH, OW, OC, device=dev, dtype=tdt)
    L.conv_igemm(xg, wf, yg, N=N, IH=H, IW=W, IC=IC, OH=OH, OW=OW, OC=OC, KH=k, KW=k, stride=s, dil=d, pad=pad, bm_hint=256)
    np.testing.assert_allclose(yg.float().cpu().numpy(), _nhwc(y.detach()).numpy(), **tol)
    # the two tile geometries must agree to accumulation-order noise
    y128 = torch.empty_like(yg)
    L.conv_igemm(xg, wf, y128, N=N, IH=H, IW=W, IC=IC, OH=OH, OW=OW, OC=OC, KH=k, KW=k, stride=s, dil=d, pad=pad, bm_hint=128)
    assert float((yg.float() - y128.float()).abs().max()) <= 2e-2
    # 224-row tiles of the 256-tile kernel (each wave row owns 112 rows)
    y7 = torch.full_like(yg, float("nan"))
    L.conv_igemm(xg, wf, y7, N=N, IH=H, IW=W, IC=IC, OH=OH, OW=OW, OC=OC, KH=k, KW=k, stride=s, dil=d, pad=pad, bm_hint=224)
    np.testing.assert_allclose(y7.float().cpu().numpy(), _nhwc(y.detach()).numpy(), **tol)
    # the 512 x 128 tile kernel (four stacked A half-tiles, all 160 KiB of LDS)
    y5 = torch.full_like(yg, float("nan"))
    L.conv_igemm(xg, wf, y5, N=N, IH=H, IW=W, IC=IC, OH=OH, OW=OW, OC=OC, KH=k, KW=k, stride=s, dil=d, pad=pad, bm_hint=259)
    np.testing.assert_allclose(y5.float().cpu().numpy(), _nhwc(y.detach()).numpy(), **tol)
    if s == 1 and IC % 128 == 0:
        dx5 = torch.full((N, H, W, IC), float("nan"), device=dev, dtype=tdt)
        L.conv_igemm(_nhwc(dy).to(dev, tdt), wt, dx5, N=N, IH=OH, IW=OW, IC=OC, OH=H, OW=W, OC=IC, KH=k, KW=k, stride=s, dil=d,
                     pad=pad, mode=1, bm_hint=259)
        np.testing.assert_allclose(dx5.float().cpu().numpy(), _nhwc(x.grad).numpy(), **tol)
    if IC % 256 == 0:                     # dgrad: the conv's IC is the GEMM's N
        dxg = torch.empty(N, H, W, IC, device=dev, dtype=tdt)
        L.conv_igemm(_nhwc(dy).to(dev, tdt), wt, dxg, N=N, IH=OH, IW=OW, IC=OC, OH=H, OW=W, OC=IC, KH=k, KW=k, stride=s, dil=d,
                     pad=pad, mode=1, bm_hint=256)
        np.testing.assert_allclose(dxg.float().cpu().numpy(), _nhwc(x.grad).numpy(), **tol)
        dx7 = torch.full_like(dxg, float("nan"))
        L.conv_igemm(_nhwc(dy).to(dev, tdt), wt, dx7, N=N, IH=OH, IW=OW, IC=OC, OH=H, OW=W, OC=IC, KH=k, KW=k, stride=s, dil=d,
                     pad=pad, mode=1, bm_hint=224)
        np.testing.assert_allclose(dx7.float().cpu().numpy(), _nhwc(x.grad).numpy(), **tol)


@pytest.mark.parametrize("geom", [(2, 132, 128, 256, 256, 3, 1, None), (2, 88, 100, 512, 256, 3, 2, None), (2, 96, 96, 512, 256, 1, 1, (36, 44)),
                                  (1, 40, 40, 256, 256, 3, 1, None)])
def test_conv_bwd_pair(geom):
    """wseg_conv_bwd_pair: a layer's data gradient and weight gradient as ONE grid (dgrad tiles first, weight-gradient tiles behind them).  dX must be
    bit-identical to the stand-alone data gradient (same tiles, same K order), dW equal to the stand-alone weight gradient up to the order of its float
    atomics, both right against F.conv2d's backward.  Cases: 3x3, dilation 2 with IC != OC, 1x1 with two row segments (the training step's two views);
    the last one is too small for the 256-tile kernel: the library must fall back to two launches."""
    from wseg_amd import _lib as L
    N, H, W, IC, OC, k, d, seg = geom
    tdt, dev = torch.bfloat16, "cuda"
    pad = d * (k // 2)
    sizes = [(H, W)] + ([seg] if seg else [])
    w = _rand((OC, IC, k, k), 2, (2.0 / (IC * k * k)) ** 0.5).to(tdt).float().requires_grad_(True)
    xs = [_rand((N, IC, h_, w_), 10 + i).to(tdt).float().requires_grad_(True) for i, (h_, w_) in enumerate(sizes)]
    dys = [_rand((N, OC, h_, w_), 20 + i).to(tdt).float() for i, (h_, w_) in enumerate(sizes)]
    for x, dy in zip(xs, dys):
        F.conv2d(x, w, None, 1, pad, d).backward(dy)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    xj = torch.cat([rows(x.detach()) for x in xs]).to(dev, tdt)
    dyj = torch.cat([rows(dy) for dy in dys]).to(dev, tdt)
    wt = torch.empty(IC, k * k, OC, device=dev, dtype=tdt)
    L.pack_weights(w.detach().permute(0, 2, 3, 1).contiguous().to(dev), None, wt, OC, k * k, IC, OC, IC, L.dtype_code(wt))
    seg2 = (seg[0], seg[1], seg[0], seg[1]) if seg else None
    geo = dict(N=N, KH=k, KW=k, stride=1, dil=d, pad=pad, seg2=seg2)
    wkw = dict(IH=H, IW=W, IC=IC, OH=H, OW=W, OC=OC, **geo)
    dx_a = torch.full((xj.shape[0], IC), float("nan"), device=dev, dtype=tdt)
    L.conv_igemm(dyj, wt, dx_a, IH=H, IW=W, IC=OC, OH=H, OW=W, OC=IC, mode=1, **geo)
    dw_a = torch.zeros(OC, k * k, IC, device=dev, dtype=torch.float32)
    L.conv_wgrad(xj, dyj, dw_a, **wkw)
    dx_b = torch.full_like(dx_a, float("nan"))
    dw_b = torch.zeros_like(dw_a)
    L.TRACK_PAIRS = True
    try:
        L.conv_igemm(dyj, wt, dx_b, IH=H, IW=W, IC=OC, OH=H, OW=W, OC=IC, mode=1, pair_wgrad=(xj, dyj, dw_b, wkw), **geo)
        fused = L.LAST_PAIR_FUSED
    finally:
        L.TRACK_PAIRS = False
    assert fused == (0 if H == 40 else 1)
    assert torch.equal(dx_a, dx_b)
    ref_dx = torch.cat([rows(x.grad) for x in xs]).numpy()
    np.testing.assert_allclose(dx_b.float().cpu().numpy(), ref_dx, rtol=2e-2, atol=2e-2)
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(OC, k * k, IC).numpy()
    scale = np.abs(ref_dw).max()
    assert np.abs(dw_b.cpu().numpy() - ref_dw).max() / scale < 1e-2
    assert float((dw_a - dw_b).abs().max()) / scale < 1e-4


def test_conv_bwd_pair_two_sources():
    """The joint grid with a two-source data gradient (a bottleneck block's `D . W_branch1 + du1 . W_branch2a`, mode 1, BN-ReLU-backward epilogue with a
    mask) and the skip conv's weight gradient: bit-identical dX to the stand-alone two-source launch, dW to the order of the atomics."""
    from wseg_amd import _lib as L
    tdt, dev = torch.bfloat16, "cuda"
    N, H, W, C1, C2, OCd = 2, 96, 100, 512, 128, 256           # D [M, 512], du1 [M, 128] -> d_t [M, 256]; weight gradient of the 256 -> 512 skip conv
    M = N * H * W
    D = _rand((M, C1), 1).to(dev, tdt)
    du1 = _rand((M, C2), 2).to(dev, tdt)
    t = _rand((M, OCd), 3).to(dev, tdt)
    mask = (_rand((M, OCd), 4) > 0).to(dev, tdt)
    wcat = _rand((OCd, 1, C1 + C2), 5, (1.0 / (C1 + C2)) ** 0.5).to(dev, tdt)
    scale = (_rand((OCd,), 6) + 1.5).to(dev)
    kw = dict(N=N, IH=H, IW=W, IC=C1, OH=H, OW=W, OC=OCd, KH=1, KW=1, mode=1, in2=du1, IC2=C2, epi=1, scale=scale, mask=mask)
    wkw = dict(N=N, IH=H, IW=W, IC=OCd, OH=H, OW=W, OC=C1, KH=1, KW=1)
    dx_a = torch.full((M, OCd), float("nan"), device=dev, dtype=tdt)
    L.conv_igemm(D, wcat, dx_a, **kw)
    dw_a = torch.zeros(C1, 1, OCd, device=dev, dtype=torch.float32)
    L.conv_wgrad(t, D, dw_a, **wkw)
    dx_b, dw_b = torch.full_like(dx_a, float("nan")), torch.zeros_like(dw_a)
    L.TRACK_PAIRS = True
    try:
        L.conv_igemm(D, wcat, dx_b, pair_wgrad=(t, D, dw_b, wkw), **kw)
        assert L.LAST_PAIR_FUSED == 1
    finally:
        L.TRACK_PAIRS = False
    assert torch.equal(dx_a, dx_b) and torch.isfinite(dx_b.float()).all()
    ref = (D.float() @ wcat[:, 0, :C1].float().t() + du1.float() @ wcat[:, 0, C1:].float().t()) * scale * (mask.float() != 0)
    np.testing.assert_allclose(dx_b.float().cpu().numpy(), ref.cpu().numpy(), rtol=2e-2, atol=3e-2)
    ref_dw = (D.float().t() @ t.float()).cpu().numpy()
    sc = np.abs(ref_dw).max()
    assert np.abs(dw_b[:, 0].cpu().numpy() - ref_dw).max() / sc < 1e-2
    assert float((dw_a - dw_b).abs().max()) / sc < 1e-4


@pytest.mark.parametrize("case", CASES256)
def test_conv256_split_bf16(case):
    """The 256-tile kernel on f32 storage with split-bf16 products (dtype WSEG_F32X3): forward with the fused BN-ReLU second output
    and a residual, data gradient with the masked-scale epilogue, on 256-row, 224-row and row-split tiles, against torch's f32 CPU
    convolution at 1e-4 (the exact-f32 kernel: 2e-5; the bf16 kernel: 2e-2)."""
    from wseg_amd import _lib as L
    N, H, W, IC, OC, k, s, d = case
    if IC % 32 != 0:
        pytest.skip("split-bf16 packs need IC % 32 == 0")
    pad = d * (k // 2)
    OH = (H + 2 * pad - d * (k - 1) - 1) // s + 1
    OW = (W + 2 * pad - d * (k - 1) - 1) // s + 1
    x = _rand((N, IC, H, W), 1).requires_grad_(True)
    w = _rand((OC, IC, k, k), 2, (2.0 / (IC * k * k)) ** 0.5).requires_grad_(True)
    dy = _rand((N, OC, OH, OW), 3)
    res = _rand((N, OC, OH, OW), 4)
    scale, shift = _rand((OC,), 5) + 1.5, _rand((OC,), 6)
    y = F.conv2d(x, w, None, s, pad, d)
    y.backward(dy)
    dev = "cuda"
    xg = _nhwc(x.detach()).to(dev)
    wm = w.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    wf32 = torch.empty(OC, k * k, IC, device=dev)
    wt32 = torch.empty(IC, k * k, OC, device=dev)
    L.pack_weights(wm, wf32, wt32, OC, k * k, IC, OC, IC, L.F32)
    wf, wt = torch.empty_like(wf32), torch.empty_like(wt32)
    L.pack_x3(wf32, wf); L.pack_x3(wt32, wt)
    tol = dict(rtol=1e-4, atol=1e-4)
    y_ref = _nhwc(y.detach() + res).numpy()
    t_ref = _nhwc(torch.relu((y.detach() + res) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))).numpy()
    for bm in (256, 224, 128):
        yg = torch.full((N, OH, OW, OC), float("nan"), device=dev)
        tg = torch.full((N, OH, OW, OC), float("nan"), device=dev)
        L.conv_igemm(xg, wf, yg, tg, N=N, IH=H, IW=W, IC=IC, OH=OH, OW=OW, OC=OC, KH=k, KW=k, stride=s, dil=d, pad=pad, bm_hint=bm,
                     r_post=_nhwc(res).to(dev), scale=scale.to(dev), shift=shift.to(dev), dtype=L.F32X3)
        np.testing.assert_allclose(yg.cpu().numpy(), y_ref, **tol)
        np.testing.assert_allclose(tg.cpu().numpy(), t_ref, **tol)
    if IC % 256 == 0:                     # dgrad: the conv's IC is the GEMM's N; BN-ReLU backward epilogue (mask = saved activation)
        mask = _rand((N, IC, H, W), 7)
        sc_in = _rand((IC,), 8) + 1.5
        dx_ref = _nhwc(x.grad * sc_in.view(1, -1, 1, 1) * (mask > 0)).numpy()
        for bm in (256, 224, 128):
            dxg = torch.full((N, H, W, IC), float("nan"), device=dev)
            L.conv_igemm(_nhwc(dy).to(dev), wt, dxg, N=N, IH=OH, IW=OW, IC=OC, OH=H, OW=W, OC=IC, KH=k, KW=k, stride=s, dil=d,
                         pad=pad, mode=1, bm_hint=bm, epi=1, scale=sc_in.to(dev), mask=_nhwc(mask).to(dev), dtype=L.F32X3)
            np.testing.assert_allclose(dxg.cpu().numpy(), dx_ref, **tol)


def test_batched_transposed_pack_bf16_matches_f32_source():
    """The two batched transposed-pack kernels (f32 master -> bf16, and bf16 mirror -> bf16 on 64x64 tiles) agree bit for bit
    on a table of layers with odd sizes."""
    from wseg_amd import _lib as L
    dev = "cuda"
    layers = [(96, 9, 40), (64, 1, 200), (130, 9, 72), (32, 1, 32)]          # (OC, T, IC)
    total = sum(o * t * i for o, t, i in layers)
    master = _rand((total,), 5).to(dev)
    mirror = master.to(torch.bfloat16)
    rows32, rows64, t32, t64, off = [], [], 0, 0, 0
    for (o, t, i) in layers:
        rows32.append([t32, off, off, o, t, i]); rows64.append([t64, off, off, o, t, i])
        t32 += ((o + 31) // 32) * ((i + 31) // 32) * t
        t64 += ((o + 63) // 64) * ((i + 63) // 64) * t
        off += o * t * i
    a = torch.zeros(total, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(total, device=dev, dtype=torch.bfloat16)
    L.pack_transposed_batch(master, a, torch.tensor(rows32, dtype=torch.int64, device=dev), len(layers), t32, L.BF16)
    L.pack_transposed_batch_bf16(mirror, b, torch.tensor(rows64, dtype=torch.int64, device=dev), len(layers), t64)
    assert torch.equal(a, b)
    o, t, i = layers[0]
    ref = mirror[:o * t * i].view(o, t, i).permute(2, 1, 0).contiguous().view(-1)
    assert torch.equal(b[:o * t * i], ref)


@pytest.mark.parametrize("bm,IC2", [(0, 192), (224, 192), (0, 64), (0, 448)])
def test_conv_two_sources(bm, IC2):
    """Two-source 1x1 (K-concatenation of a bottleneck's skip conv and last conv): out = in . W[:,0,:] + in2 . W[:,1,:] with the
    fused BN-ReLU-dropout second output, two row segments, against the two separate convolutions on the CPU."""
    from wseg_amd import _lib as L
    tdt = torch.bfloat16
    N, IC, OC = 2, 192, 512
    (H1, W1), (H2, W2) = (21, 17), (8, 9)
    xs = [[_rand((N, c, h, w), 1 + 2 * s_ + v).to(tdt).float() for (h, w) in ((H1, W1), (H2, W2))] for v, s_, c in ((0, 0, IC), (1, 5, IC2))]
    w1 = _rand((OC, IC, 1, 1), 30, (1.0 / IC) ** 0.5).to(tdt).float()
    w2 = _rand((OC, IC2, 1, 1), 31, (1.0 / IC2) ** 0.5).to(tdt).float()
    scale, shift = _rand((OC,), 6) + 1.5, _rand((OC,), 7)
    drop = (torch.rand(2 * N, OC, generator=torch.Generator().manual_seed(8)) > 0.5).float() * 2
    rows = lambda t: _nhwc(t).reshape(-1, t.shape[1])
    ref = torch.cat([rows(F.conv2d(a, w1) + F.conv2d(b, w2)) for a, b in zip(xs[0], xs[1])])
    img = torch.cat([torch.arange(N).repeat_interleave(H1 * W1), N + torch.arange(N).repeat_interleave(H2 * W2)])
    act = F.relu(ref * scale + shift) * drop[img]
    dev = "cuda"
    a = torch.cat([rows(t) for t in xs[0]]).to(dev, tdt)
    b = torch.zeros(a.shape[0], IC2 + 64, device=dev, dtype=tdt)         # second source with a row stride
    b[:, :IC2] = torch.cat([rows(t) for t in xs[1]]).to(dev, tdt)
    wcat = torch.cat([w1.reshape(OC, 1, IC), w2.reshape(OC, 1, IC2)], dim=2).contiguous().to(dev, tdt)   # rows [W1[oc] | W2[oc]]
    out = torch.full((a.shape[0], OC), float("nan"), device=dev, dtype=tdt)
    out2 = torch.full_like(out, float("nan"))
    L.conv_igemm(a, wcat, out, out2, N=N, IH=H1, IW=W1, IC=IC, OH=H1, OW=W1, OC=OC, KH=1, KW=1, seg2=(H2, W2, H2, W2),
                 in2=b, IC2=IC2, ld_in2=IC2 + 64, scale=scale.to(dev), shift=shift.to(dev), drop=drop.to(dev), bm_hint=bm)
    tol = dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), **tol)
    np.testing.assert_allclose(out2.float().cpu().numpy(), act.numpy(), rtol=2e-2, atol=4e-2)
    with pytest.raises(RuntimeError):                                    # only same-size stride-1 forms exist (rejected before any launch)
        L.conv_igemm(a, wcat, out, N=N, IH=H1, IW=W1, IC=IC, OH=(H1 + 1) // 2, OW=(W1 + 1) // 2, OC=OC, KH=1, KW=1, stride=2,
                     seg2=(H2, W2, (H2 + 1) // 2, (W2 + 1) // 2), in2=b, IC2=IC2, ld_in2=IC2 + 64)


@pytest.mark.parametrize("dil,IC2", [(1, 128), (2, 320)])
def test_conv_two_sources_3x3(dil, IC2):
    """Residual-block form of the two-source product: a same-size 3x3 (dilated) convolution plus a 1x1 on a second input, forward
    (mode 0) and as the sum of the two data gradients (mode 1 on the transposed packs), two row segments."""
    from wseg_amd import _lib as L
    tdt = torch.bfloat16
    N, IC, OC, k = 2, 128, 256, 3
    (H1, W1), (H2, W2) = (19, 16), (7, 10)
    pad = dil
    x1 = [_rand((N, IC, h, w), 1 + i).to(tdt).float() for i, (h, w) in enumerate(((H1, W1), (H2, W2)))]
    x2 = [_rand((N, IC2, h, w), 11 + i).to(tdt).float() for i, (h, w) in enumerate(((H1, W1), (H2, W2)))]
    w1 = _rand((OC, IC, k, k), 30, (1.0 / (9 * IC)) ** 0.5).to(tdt).float()
    w2 = _rand((OC, IC2, 1, 1), 31, (1.0 / IC2) ** 0.5).to(tdt).float()
    rows = lambda t: _nhwc(t).reshape(-1, t.shape[1])
    ref = torch.cat([rows(F.conv2d(a, w1, None, 1, pad, dil) + F.conv2d(b, w2)) for a, b in zip(x1, x2)])
    dev = "cuda"
    a = torch.cat([rows(t) for t in x1]).to(dev, tdt)
    b = torch.cat([rows(t) for t in x2]).to(dev, tdt)
    wcat = torch.cat([w1.permute(0, 2, 3, 1).reshape(OC, 9 * IC), w2.reshape(OC, IC2)], dim=1).contiguous().to(dev, tdt)
    out = torch.full((a.shape[0], OC), float("nan"), device=dev, dtype=tdt)
    L.conv_igemm(a, wcat, out, N=N, IH=H1, IW=W1, IC=IC, OH=H1, OW=W1, OC=OC, KH=k, KW=k, dil=dil, pad=pad, seg2=(H2, W2, H2, W2), in2=b, IC2=IC2)
    tol = dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), **tol)
    # mode 1: dX = dgrad_3x3(dY1; V1) + dY2 . V2 for convolutions V1 [OCv][IC_v][3][3] (input = this test's output side) and V2 1x1
    OCv = IC                                           # dY1 has OCv channels, dX has OC channels
    v1 = _rand((OCv, OC, k, k), 40, (1.0 / (9 * OCv)) ** 0.5).to(tdt).float()     # conv OC -> OCv
    v2 = _rand((IC2, OC, 1, 1), 41, (1.0 / IC2) ** 0.5).to(tdt).float()           # conv OC -> IC2
    refd = torch.cat([rows(F.conv_transpose2d(dy1, v1, None, 1, pad, 0, 1, dil) + F.conv_transpose2d(dy2, v2)) for dy1, dy2 in zip(x1, x2)])
    wt = torch.cat([v1.permute(1, 2, 3, 0).reshape(OC, 9 * OCv), v2.reshape(IC2, OC).t()], dim=1).contiguous().to(dev, tdt)   # [OC][9*OCv + IC2]
    dx = torch.full((a.shape[0], OC), float("nan"), device=dev, dtype=tdt)
    L.conv_igemm(a, wt, dx, N=N, IH=H1, IW=W1, IC=OCv, OH=H1, OW=W1, OC=OC, KH=k, KW=k, dil=dil, pad=pad, mode=1, seg2=(H2, W2, H2, W2), in2=b, IC2=IC2)
    np.testing.assert_allclose(dx.float().cpu().numpy(), refd.numpy(), **tol)
