"""Fused training-step loss on the HIP kernels (contrast_train.py:138-398).

Drives the kernels of csrc/loss.hip between the two `Engine.run_forward(..., lowres=True)` calls and
the two `Engine.run_backward` calls: every loss value AND every gradient down to the stride-8 maps is
computed by hand-written kernels; torch is used for allocation, the RCCL exchange and the final
combination of a handful of device scalars.  No autograd graph is built.

Global-batch semantics under data parallelism: the per-class top-32 prototype
candidates (value + feature row) of every rank are all-gathered and merged, so every rank holds the
prototypes the reference would compute over the whole batch (one all-gather for both views); hard-pixel sampling (contrast_train.py
:302-331) exchanges one {label, similarity, random key} record per pixel, and every rank finds the same
global per-class order statistics (csrc/loss.hip intra_weights_global).
"""
import os

import torch
import torch.distributed as dist

from . import _lib as L
from .engine import HEAD_LD

_TIE_CACHE = {}


def _side_streams(eng, dev):
    st = getattr(eng, "_side_streams", None)
    if st is None or st[0].device != dev:
        st = eng._side_streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
    return st


def _prefix_stream(eng, dev):
    """Stream of the next batch's frozen prefix (step(lookahead=...)).  An ordinary stream: a high-priority or CU-masked one
    (hipExtStreamCreateWithCUMask, to keep a few CUs free for the loss kernels) slowed EVERY conv launch of the step (36.1 -> 45-49 ms)."""
    st = getattr(eng, "_prefix_stream", None)
    if st is None or st.device != dev:
        st = eng._prefix_stream = torch.cuda.Stream(dev)
    return st


def _aux_stream(eng, dev):
    st = getattr(eng, "_aux_stream", None)
    if st is None or st.device != dev:
        st = eng._aux_stream = torch.cuda.Stream(dev)
    return st


def cpu_tie_pattern(P, k, device=None):
    """Q5: the index set torch.topk returns for a fully tied row of length P on the CPU library the
    reference's CPU path uses.  Data independent; computed once per (P, k) (and uploaded once per device)."""
    key = (P, k)
    if key not in _TIE_CACHE:
        _TIE_CACHE[key] = torch.topk(torch.full((1, P), 0.2), k, dim=-1)[1][0].to(torch.int32)
    if device is None:
        return _TIE_CACHE[key]
    dkey = (P, k, str(device))
    if dkey not in _TIE_CACHE:
        _TIE_CACHE[dkey] = _TIE_CACHE[key].to(device)
    return _TIE_CACHE[dkey]


def _f32(*shape, dev):
    return torch.empty(shape, device=dev, dtype=torch.float32)


class _View:
    pass


_WVEC = {}


def _adjoint_ones(h, S, dev):
    """wvec[y] = total weight of low-res row y in the align_corners upsample h -> S (the GAP gradient pattern)."""
    key = (h, S, str(dev))
    if key not in _WVEC:
        wv = torch.empty(h, device=dev, dtype=torch.float32)
        L.resize_adjoint_ones(wv, h, S)
        _WVEC[key] = wv
    return _WVEC[key]


def _maps_forward(v, label20, acc, N):
    """Plane statistics, cls + rvmin losses, max-norm + 128x128 resize for one view — all evaluated on the fly from
    the stride-8 maps (csrc/maps.hip): the [N,21,S,S] upsampled maps of the reference are never materialised."""
    dev = v.cam_low.device
    S, h, w = v.S, v.h, v.w
    npix = S * S
    v.st_cam = _f32(N * 21, 6, dev=dev)
    v.st_rv = _f32(N * 21, 6, dev=dev)
    L.up_plane_stats(v.cam_low, v.st_cam, N * 21, h, w, S)
    L.up_plane_stats(v.rvd, v.st_rv, N * 21, h, w, S, label20)        # (max/min of labelled planes only; no GAP on this map)
    v.bias = _f32(N * 21, dev=dev)
    L.cls_loss(v.st_cam, label20, acc[0:1], v.bias, N, npix, 0.5)
    v.q = _f32(N, npix, dev=dev)
    v.argc = torch.empty(N, npix, device=dev, dtype=torch.uint8)
    L.up_rvmin_values(v.rvd, label20, v.q, v.argc, N, h, w, S)
    v.k_min = npix // 4
    v.res_min = _f32(N, 4, dev=dev)
    ws = torch.empty(L.select_workspace_bytes(N), device=dev, dtype=torch.uint8)
    L.select_kth(v.q, N, npix, v.k_min, False, False, True, v.res_min, ws)
    L.select_finish(v.res_min, N, v.k_min, True, 0.5 / (v.k_min * N), acc[1:2])
    v.c = _f32(N, 21, 128, 128, dev=dev)
    v.r = _f32(N, 21, 128, 128, dev=dev)
    L.up_norm_resize_forward(v.cam_low, v.st_cam, label20, v.c, N, h, w, S, 128)
    L.up_norm_resize_forward(v.rvd, v.st_rv, label20, v.r, N, h, w, S, 128)


def _maps_backward_cam(v, label20, N):
    """d(loss)/d(cam_low): cls + ER terms — needs er_ecr_prep's Gc only, not the ECR selection"""
    dev = v.cam_low.device
    S, h, w = v.S, v.h, v.w
    wy, wx = _adjoint_ones(h, S, dev), _adjoint_ones(w, S, dev)
    v.d_cam_low = _f32(N, 21, h, w, dev=dev)
    L.up_maps_backward(v.Gc, v.cam_low, v.st_cam, label20, v.bias, wy, wx, None, None, None, 0, 0.0, v.d_cam_low, N, h, w, S, 128)


def _maps_backward_rv(v, label20, N):
    """d(loss)/d(cam_rv_down): min-pool + ER + ECR terms"""
    dev = v.cam_low.device
    S, h, w = v.S, v.h, v.w
    v.d_rvd = _f32(N, 21, h, w, dev=dev)
    L.up_maps_backward(v.Gr, v.rvd, v.st_rv, label20, None, None, None, v.q, v.argc, v.res_min, v.k_min, 0.5 / (v.k_min * N),
                       v.d_rvd, N, h, w, S, 128)


_CAND_K = 256 // 8                                           # top-32 per class (contrast_train.py:202)
_CAND_L = 21 * _CAND_K * 129 + 32                            # one view's candidate block: [21][K] values | [21][K][128] features | [21] flags (+pad)


def _candidates(v, label20, bg_threshold, tie_idx, N, block):
    """Pseudo-labels and this rank's per-class top-32 prototype candidates of one view, written into `block` (a
    _CAND_L-float slice of the exchange buffer)."""
    dev = v.cam_low.device
    P = N * 256
    K = _CAND_K
    v.F = _f32(P, 128, dev=dev)
    L.rows_resize_forward(v.head, HEAD_LD, v.F, N, v.h, v.w, 16, 16)
    if (v.h, v.w) == (16, 16):
        R = v.rvd
    else:
        R = _f32(N, 21, 16, 16, dev=dev)
        L.resize_planar_fwd(v.rvd, R, N * 21, v.h, v.w, 16, 16, True)
    v.y = torch.empty(P, device=dev, dtype=torch.int32)
    v.ncam = _f32(N, 21, 256, dev=dev)
    L.pseudo_label(R, label20, bg_threshold, v.y, v.ncam, N, 256)
    cv, cf = block[:21 * K], block[21 * K:21 * K * 129]
    cc = block[21 * K * 129:21 * K * 129 + 21].view(torch.int32)
    L.proto_candidates(v.ncam, v.F, tie_idx, cv, cf, cc, N, 256, K)


def _merge_prototypes(v, gathered, view_idx, world):
    """Global-batch prototypes of one view from the gathered candidate blocks [world][2][_CAND_L] (every rank: same result)."""
    K = _CAND_K
    base = gathered.view(-1)[view_idx * _CAND_L:]
    v.protos = _f32(21, 128, dev=gathered.device)
    L.proto_merge(base, base[21 * K:], base[21 * K * 129:].view(torch.int32), v.protos, world, K, 2 * _CAND_L)


def _random_keys(P, rank, view_idx, dev):
    """Per-pixel random keys of the hard-pixel sampling (the 'random half' = the smallest keys of a class).  Normally
    torch.rand; WSEG_INTRA_KEY_SEED=<int> makes them a fixed function of the GLOBAL pixel index, so that a data-parallel run
    and a single-process run over the same global batch pick the same pixels (tests/test_gpu_ddp_equivalence.py)."""
    seed = os.environ.get("WSEG_INTRA_KEY_SEED")
    if seed is None:
        return torch.rand(P, device=dev)
    g = torch.arange(rank * P, (rank + 1) * P, device=dev, dtype=torch.int64)
    h = (g * 2654435761 + int(seed) * 40503 + view_idx * 97) % 16777213
    return (h * 48271 % 16777213).float() / 16777213.0


def _rand_flags(y_dev, rng, P):
    """RNG-parity mode: the reference's host draws (contrast_train.py:291/:316), as per-pixel flags."""
    y = y_dev.cpu()
    for _ in range(P):
        rng.sample(range(21), 10)
    flags = torch.zeros(P, dtype=torch.uint8)
    for cls in torch.unique(y).tolist():
        members = (y == cls).nonzero(as_tuple=True)[0]
        n_c = members.numel()
        if n_c < 2:
            continue
        flags[members[torch.tensor(rng.sample(range(n_c), n_c // 2), dtype=torch.long)]] = 1
    return flags.to(y_dev.device)


def step(model, img1, img2, label20, bg_threshold=0.20, rng=None, rng_parity=False, bg_topk_idx=None, zero_grads=False, prefix=None, lookahead=None):
    """Forward both views, all losses, backward into the engine's flat gradient buffer (accumulating; zero_grads=True clears
    the buffer first — on a side stream during the loss phase, where it costs nothing).  Returns the 8 logged scalars
    (device tensors).
    prefix: Engine.run_prefix([img1, img2]) when an earlier call has computed it.  lookahead: {"img1": the NEXT batch's images}; this call adds
    "img2" and "prefix" for them: the frozen part of the next forward pass (conv1a, b2*: 2.2 ms of conv work that depends on no trainable weight)
    runs on its own stream between this step's forward and backward passes, where ~2 ms of small loss kernels leave the chip mostly idle."""
    eng = model._engine
    dev = img1.device
    N = img1.shape[0]
    from .train import dist_state
    world, distributed = dist_state()
    label20 = label20.to(dev).float().contiguous()
    eng.ensure_flat(dev)
    eng.attach_grads()
    defer = os.environ.get("WSEG_DEFER_PACKS", "1") != "0"      # (0: A/B switch — packs and memset before the forward pass)
    from .engine import DT_OF
    use_streams = os.environ.get("WSEG_STREAMS", "1") != "0"
    eng.ensure_packs(dev, DT_OF[model.precision], defer_wt=defer, late_stream=_aux_stream(eng, dev) if (defer and use_streams) else None)
    if zero_grads and not defer:
        eng.flat_g.zero_()
    acc = torch.zeros(8, device=dev, dtype=torch.float32)   # [cls1+cls2, (rvmin1+rvmin2)/2, er_sum, ecr, cross, cross2, intra]
    # The two views are independent until ER/ECR: run each on its own HIP stream so the small 128x128 view's
    # launches (which cannot fill 256 CUs) overlap with the 448x448 view's.
    main = torch.cuda.current_stream(dev)
    side = _side_streams(eng, dev) if use_streams else (main, main)
    # Both views go through the network in ONE batched pass (two row segments per launch); the per-view map
    # losses then run on their own HIP streams.
    outs, ctx = eng.run_forward([img1, img2], save=True, lowres=True, prefix=prefix)
    fork = main.record_event()
    pst = None
    if lookahead is not None:
        pst = _prefix_stream(eng, dev)
        n1 = lookahead["img1"]
        # the two tensors that cross from the prefix stream into the next step are allocated HERE, on the caller's stream (its allocator pool: they are
        # written on the prefix stream, which the caller waits for before its backward pass, and freed in the caller's program order)
        n2 = torch.empty(n1.shape[0], n1.shape[1], img2.shape[-2], img2.shape[-1], device=dev, dtype=torch.float32)
        rows, ch = eng.prefix_out_shape([n1, n2])
        t_next = torch.empty(rows, ch, device=dev, dtype=L.TORCH_DTYPE[DT_OF[model.precision]])
        pst.wait_event(fork)
        with torch.cuda.stream(pst):
            L.resize_planar_fwd(n1, n2, n1.shape[0] * n1.shape[1], n1.shape[2], n1.shape[3], n2.shape[2], n2.shape[3], True)
            lookahead["img2"], lookahead["prefix"] = n2, eng.run_prefix([n1, n2], out=t_next)
    # Backward-only preparation — the transposed weight packs (420 MB of traffic) and the gradient memset (420 MB) — on a third
    # stream behind the forward pass: it runs while the loss phase's small kernels leave the chip's bandwidth idle.
    aux = _aux_stream(eng, dev) if use_streams else main
    aux.wait_event(fork)
    with torch.cuda.stream(aux):
        eng.finish_packs()
        if zero_grads and defer:
            eng.flat_g.zero_()
    P = N * 256
    tie_idx = (bg_topk_idx.to(device=dev, dtype=torch.int32) if bg_topk_idx is not None else cpu_tie_pattern(P, 32, dev))
    views = []
    for img, (cam_low, rvd, _fp, head), vw in zip((img1, img2), outs, ctx["views"]):
        v = _View()
        v.S, v.h, v.w, v.off = img.shape[2], vw["h"], vw["w"], vw["off"]
        v.cam_low, v.rvd, v.head = cam_low, rvd, head
        views.append(v)
    v1, v2 = views
    # The loss phase is ~60 small dependent kernels; its critical path is what the step waits for.  HIP multiplexes its streams onto FOUR hardware
    # queues (raising GPU_MAX_HW_QUEUES slows every conv launch: measured 37.0 -> 41-42 ms / step), one of which carries the weight packs (aux), so the
    # independent branches are laid out on three lanes:
    #   side[0]: map losses of view 1  M(1) -> [M(2)] -> ER / ECR chain -> map backward of view 1
    #   side[1]: map losses of view 2  M(2) ------------------------^ -> hard-pixel weights of view 2 -> map backward of view 2
    #   main:    pseudo-labels + prototype candidates of both views (they need only the forward outputs) -> prototypes -> record pass
    #            -> [all-gather] -> hard-pixel weights of view 1 -> fused NCE -> [side 0, 1] -> head gradient -> backward pass
    cand = _f32(2, _CAND_L, dev=dev)
    for vi, v in enumerate(views):
        side[vi].wait_event(fork)
        with torch.cuda.stream(side[vi]):
            _maps_forward(v, label20, acc, N)
    for vi, v in enumerate(views):
        _candidates(v, label20, bg_threshold, tie_idx, N, cand[vi])
    gathered = cand
    if distributed:                                         # global-batch prototypes: ONE all-gather of both views' candidates (694 KB)
        gathered = _f32(world, 2, _CAND_L, dev=dev)
        dist.all_gather_into_tensor(gathered.view(world * 2, _CAND_L), cand)
    for vi, v in enumerate(views):
        _merge_prototypes(v, gathered, vi, world)
    # ---- ER + ECR on the 128x128 maps (both directions of the ECR top-k in ONE 2N-row selection), on side[0] behind both views' map losses
    npix = 128 * 128
    er_coef = 1.0 / (N * 20 * npix)
    est = side[0]
    est.wait_stream(side[1])
    with torch.cuda.stream(est):
        for v in views:
            v.Gc = _f32(N, 21, 128, 128, dev=dev)
        dlt = _f32(2 * N, 21 * npix, dev=dev)
        L.er_ecr_prep(v1.c, v2.c, v1.r, v2.r, v1.Gc, v2.Gc, dlt[:N], dlt[N:], acc[2:3], N, npix, er_coef)
        prep_done = est.record_event()
        K_ecr = int(21 * npix * 0.2)
        ws = torch.empty(L.select_workspace_bytes(2 * N), device=dev, dtype=torch.uint8)
        res = _f32(2 * N, 4, dev=dev)
        L.select_kth(dlt, 2 * N, 21 * npix, K_ecr, True, True, False, res, ws)
        L.select_finish(res, 2 * N, K_ecr, False, 1.0 / (N * K_ecr), acc[3:4])
        Gr = _f32(2 * N, 21, 128, 128, dev=dev)
        L.ecr_backward(dlt, res, Gr, 2 * N, 21 * npix, K_ecr, 1.0 / (N * K_ecr))
        v1.Gr, v2.Gr = Gr[:N], Gr[N:]
        ecr_done = est.record_event()
        _maps_backward_rv(v1, label20, N)
    side[1].wait_event(prep_done)                           # both views' cam-map backward beside the ECR selection (side[1] is idle after M(2))
    with torch.cuda.stream(side[1]):
        _maps_backward_cam(v1, label20, N)
        _maps_backward_cam(v2, label20, N)
    # ---- pixel-to-prototype similarities + hard-pixel records (main stream, concurrently with the ER / ECR chain)
    # (the radix-select kernel is also the faster one on a single rank — 47 vs 118 us; the sort-based kernel remains the
    #  RNG-parity path, which replays the reference's host random stream)
    global_intra = distributed or (not rng_parity and os.environ.get("WSEG_INTRA_GLOBAL", "1") == "1")
    rank = dist.get_rank() if distributed else 0
    # record pass (similarities that only RANK pixels): exact-f32 MFMA in fp32 mode, split-bf16 products in the bf16 / bf16x3 modes
    nce_x3 = model.precision != "fp32" and os.environ.get("WSEG_NCE_X3", "1") != "0"
    # ONE launch for both views: per-pixel records {label, similarity to the pixel's own-class prototype, random key} straight from
    # the raw features (csrc/loss.hip nce_records): the inputs of the hard-pixel sampling.  Over the GLOBAL batch under data
    # parallelism (the reference samples on the gathered batch): the records (96 KB per rank for both views) are
    # all-gathered, every rank finds the same global per-class order statistics and keeps the weights of its own pixels, scaled
    # by `world` because the gradient all-reduce averages.
    rec = _f32(2, 3, P, dev=dev)
    for vi, v in enumerate(views):
        v.rkey = _random_keys(P, rank, vi, dev) if global_intra else None
    L.nce_records([dict(F=v.F, p_own=v.protos, y_own=v.y, rkey=v.rkey, rec=rec[vi]) for vi, v in enumerate(views)], P, split_bf16=nce_x3)
    grec = rec
    if global_intra and distributed:                        # (collectives stay on the main stream, in program order)
        grec = _f32(world, 2, 3, P, dev=dev)
        dist.all_gather_into_tensor(grec.view(world * 6, P), rec.view(6, P))
    elif rng_parity and not global_intra:
        for vi, v in enumerate(views):                     # view 1 fully before view 2 (RNG order of the reference)
            v.w_intra = _f32(P, dev=dev)
            L.intra_weights(v.y, rec[vi, 1], None, _rand_flags(v.y, rng, P), v.w_intra, P, ld_s=1)
    # hard-pixel weights (a single-workgroup kernel per view): view 1 here, view 2 on side[1] in front of its map backward
    def weights_of(vi, v):
        if global_intra:
            v.w_intra = _f32(P, dev=dev)
            L.intra_weights_global(grec.view(-1)[vi * 3 * P:], v.w_intra, P, world, rank, float(world), 6 * P)
        elif not rng_parity:
            v.w_intra = _f32(P, dev=dev)
            L.intra_weights(v.y, rec[vi, 1], torch.rand(P, device=dev), None, v.w_intra, P, ld_s=1)

    fork2 = main.record_event()
    weights_of(0, v1)
    side[1].wait_event(fork2)
    with torch.cuda.stream(side[1]):
        weights_of(1, v2)
        w2_done = side[1].record_event()
        side[1].wait_event(ecr_done)
        _maps_backward_rv(v2, label20, N)
    main.wait_event(w2_done)
    # similarities, the three InfoNCE terms and dF of BOTH views in one launch (csrc/loss.hip nce_fused): features read once, only dF written
    for v in views:
        v.dF = _f32(P, 128, dev=dev)
    L.nce_fused([dict(F=v.F, p_own=v.protos, p_oth=o.protos, y_own=v.y, y_oth=o.y, w_intra=v.w_intra, dF=v.dF) for v, o in ((v1, v2), (v2, v1))],
                P, 0.1 / (2 * P), 0.05, acc[4:7])
    # ---- into the network: one batched backward over both views
    d_head = torch.empty_like(ctx["head"])
    main.wait_stream(side[0])
    main.wait_stream(side[1])
    for v in views:
        L.head_grad_fused(v.dF, v.d_cam_low, v.head, d_head[v.off:], HEAD_LD, N, v.h, v.w, 16, 16)
    main.wait_stream(aux)
    if pst is not None:
        main.wait_stream(pst)                               # (the heavy backward launches must not share the chip with the prefix tiles)
    eng.run_backward(ctx, [(None, v.d_rvd, None, None) for v in views], d_head_rows=d_head)
    if eng.capture_ctx:                                     # tests: pseudo-labels, prototypes, hard-pixel weights of both views
        eng.last_loss_views = views
    out8 = torch.empty(8, device=dev, dtype=torch.float32)
    L.loss_finish(acc, er_coef, out8)
    return dict(zip(("loss", "loss_cls", "loss_er", "loss_ecr", "loss_nce", "loss_intra_nce", "loss_cross_nce", "loss_cross_nce2"), out8.unbind(0)))
