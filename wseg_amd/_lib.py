"""ctypes binding of libwseg_hip.so (the C ABI of include/wseg_hip.h).

There is deliberately NO fallback: if the library is missing the import raises, and every
wrapper raises RuntimeError(wseg_last_error()) on a non-zero status.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WSEG_LIB") or os.path.join(_HERE, "libwseg_hip.so")   # (WSEG_LIB: a probe build kept beside the product, development only)

F32, BF16, F32X3 = 0, 1, 2      # F32X3: f32 tensors, conv / wgrad products as split-bf16 (hi.hi + lo.hi + hi.lo)
PROFILE_WGRAD = None
PROFILE = None        # set to a list by bench.py to collect (start_event, end_event, flops) per conv launch
PROFILE_STRIDE = 1    # event-bracket launch i of step s only when (i + s) % stride == 0: an event pair costs ~10 us of
_profile_idx = 0      # stream time, 81 pairs per step would slow the timed region by ~2 %
_profile_phase = 0


def profile_begin_step(step):
    """bench.py calls this at the start of every timed step: launch indices restart, the sampling phase rotates."""
    global _profile_idx, _profile_phase
    _profile_idx, _profile_phase = 0, step % max(1, PROFILE_STRIDE)


def _profile_sample():
    global _profile_idx
    i = _profile_idx
    _profile_idx += 1
    return (i + _profile_phase) % max(1, PROFILE_STRIDE) == 0, i
TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, F32X3: torch.float32}


class ConvDesc(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("w", C.c_void_p), ("out", C.c_void_p), ("out2", C.c_void_p),
                ("r_pre", C.c_void_p), ("r_post", C.c_void_p), ("mask", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("drop", C.c_void_p),
                ("N", C.c_int32), ("IH", C.c_int32), ("IW", C.c_int32), ("IC", C.c_int32), ("ld_in", C.c_int32),
                ("OH", C.c_int32), ("OW", C.c_int32), ("OC", C.c_int32), ("ld_out", C.c_int32), ("ld_out2", C.c_int32),
                ("ld_rpre", C.c_int32), ("ld_rpost", C.c_int32), ("ld_mask", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("dil", C.c_int32), ("pad", C.c_int32),
                ("mode", C.c_int32), ("epi", C.c_int32), ("dtype", C.c_int32), ("relu_out2", C.c_int32),
                ("relu_lt", C.c_int32), ("bm_hint", C.c_int32),
                ("IH2", C.c_int32), ("IW2", C.c_int32), ("OH2", C.c_int32), ("OW2", C.c_int32),
                ("in2", C.c_void_p), ("ld_in2", C.c_int32), ("IC2", C.c_int32), ("w_rows", C.c_int32)]


class WgradDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p),
                ("N", C.c_int32), ("IH", C.c_int32), ("IW", C.c_int32), ("IC", C.c_int32), ("ld_x", C.c_int32),
                ("OH", C.c_int32), ("OW", C.c_int32), ("OC", C.c_int32), ("ld_dy", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("dil", C.c_int32), ("pad", C.c_int32),
                ("dtype", C.c_int32), ("split_k", C.c_int32), ("IC_dw", C.c_int32), ("OC_dw", C.c_int32),
                ("tile_hint", C.c_int32),
                ("IH2", C.c_int32), ("IW2", C.c_int32), ("OH2", C.c_int32), ("OW2", C.c_int32), ("dw_rot", C.c_int32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(wseg_amd has no CPU/eager fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.wseg_last_error.restype = C.c_char_p
    lib.wseg_version.restype = C.c_int
    lib.wseg_sizeof_conv_desc.restype = C.c_size_t
    lib.wseg_sizeof_wgrad_desc.restype = C.c_size_t
    if lib.wseg_sizeof_conv_desc() != C.sizeof(ConvDesc) or lib.wseg_sizeof_wgrad_desc() != C.sizeof(WgradDesc):
        raise ImportError(f"{LIB_PATH} was built from another include/wseg_hip.h (descriptor sizes "
                          f"{lib.wseg_sizeof_conv_desc()}/{lib.wseg_sizeof_wgrad_desc()} vs {C.sizeof(ConvDesc)}/{C.sizeof(WgradDesc)}): rebuild it")
    return lib


lib = _load()


def _ptr(t):
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed ({status}): {lib.wseg_last_error().decode()}")


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


TRACK_PAIRS, LAST_PAIR_FUSED = False, None        # tests: whether the last pair_wgrad launch was ONE grid


def conv_igemm(inp, w, out=None, out2=None, *, N, IH, IW, IC, OH, OW, OC, KH, KW, stride=1, dil=1, pad=0,
               mode=0, epi=0, r_pre=None, r_post=None, mask=None, scale=None, shift=None, drop=None,
               ld_in=None, ld_out=None, ld_out2=None, ld_rpre=None, ld_rpost=None, ld_mask=None, relu_out2=1,
               relu_lt=0, bm_hint=0, seg2=None, in2=None, ld_in2=None, IC2=0, dtype=None, pair_wgrad=None, w_rows=0):
    """w_rows: rows of the weight pack when it is zero-padded beyond OC (wseg_conv_desc.w_rows).  pair_wgrad: (x, dy, dw, kwargs of conv_wgrad) — a weight gradient launched in the SAME grid as this data gradient (wseg_conv_bwd_pair; the
    library falls back to two launches when the pair does not qualify)."""
    d = ConvDesc()
    d.inp, d.w, d.out, d.out2 = _ptr(inp), _ptr(w), _ptr(out), _ptr(out2)
    d.r_pre, d.r_post, d.mask = _ptr(r_pre), _ptr(r_post), _ptr(mask)
    d.scale, d.shift, d.drop = _ptr(scale), _ptr(shift), _ptr(drop)
    d.N, d.IH, d.IW, d.IC, d.ld_in = N, IH, IW, IC, ld_in or IC
    d.OH, d.OW, d.OC, d.ld_out, d.ld_out2 = OH, OW, OC, ld_out or OC, ld_out2 or OC
    d.ld_rpre, d.ld_rpost, d.ld_mask = ld_rpre or OC, ld_rpost or OC, ld_mask or OC
    d.KH, d.KW, d.stride, d.dil, d.pad = KH, KW, stride, dil, pad
    d.mode, d.epi, d.dtype, d.relu_out2, d.relu_lt, d.bm_hint = mode, epi, (dtype_code(inp) if dtype is None else dtype), relu_out2, relu_lt, bm_hint
    if seg2 is not None:                         # (IH2, IW2, OH2, OW2): second row segment, same N
        d.IH2, d.IW2, d.OH2, d.OW2 = seg2
    if in2 is not None:                          # two sources: w = [OC][KH*KW*IC + IC2]
        d.in2, d.IC2, d.ld_in2 = _ptr(in2), IC2, ld_in2 or (IC2 or IC)
    krow = KH * KW * IC + ((IC2 or IC) if in2 is not None else 0)
    d.w_rows = w_rows
    if w.numel() < max(OC, w_rows) * krow:       # (raw pointers beyond this line: a short weight buffer would be read out of bounds)
        raise RuntimeError(f"conv_igemm: weight buffer has {w.numel()} elements, the launch reads {max(OC, w_rows)} x {krow}")
    sampled, launch_idx = _profile_sample() if PROFILE is not None else (False, 0)
    if sampled:                                  # bench.py: HIP events on the launch stream around this launch
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    wflops = 0.0
    if pair_wgrad is not None:
        wx, wdy, wdw, wkw = pair_wgrad
        wd = _wgrad_desc(wx, wdy, wdw, **wkw)
        global LAST_PAIR_FUSED
        LAST_PAIR_FUSED = lib.wseg_conv_bwd_pair_fuses(C.byref(d), C.byref(wd)) if TRACK_PAIRS else None
        check(lib.wseg_conv_bwd_pair(C.byref(d), C.byref(wd), C.c_void_p(stream_ptr())), "wseg_conv_bwd_pair")
        wpix = wkw["N"] * wkw["OH"] * wkw["OW"] + (wkw["N"] * wkw["seg2"][2] * wkw["seg2"][3] if wkw.get("seg2") is not None else 0)
        wflops = 2.0 * wpix * wkw["IC"] * wkw["OC"] * wkw["KH"] * wkw["KW"]
    else:
        check(lib.wseg_conv_igemm(C.byref(d), C.c_void_p(stream_ptr())), "wseg_conv_igemm")
    if sampled:
        ev1.record()
        pix = N * (OH * OW if mode == 0 else IH * IW)        # algorithmic: the conv's output pixels
        if seg2 is not None:
            pix += N * (seg2[2] * seg2[3] if mode == 0 else seg2[0] * seg2[1])
        kin = IC * KH * KW + ((IC2 or IC) if in2 is not None else 0)
        PROFILE.append((ev0, ev1, 2.0 * pix * kin * OC + wflops,
                        f"{'fwd' if mode == 0 else ('dgrad+wgrad' if pair_wgrad is not None else 'dgrad')} {IC}{('+%d' % (IC2 or IC)) if in2 is not None else ''}->{OC} k{KH} s{stride} d{dil} {OH}x{OW}", launch_idx))


def _wgrad_desc(x, dy, dw, *, N, IH, IW, IC, OH, OW, OC, KH, KW, stride=1, dil=1, pad=0,
                ld_x=None, ld_dy=None, split_k=0, IC_dw=None, OC_dw=None, tile_hint=0, seg2=None, dtype=None, dw_rot=0):
    d = WgradDesc()
    d.x, d.dy, d.dw = _ptr(x), _ptr(dy), _ptr(dw)
    d.N, d.IH, d.IW, d.IC, d.ld_x = N, IH, IW, IC, ld_x or IC
    d.OH, d.OW, d.OC, d.ld_dy = OH, OW, OC, ld_dy or OC
    d.KH, d.KW, d.stride, d.dil, d.pad = KH, KW, stride, dil, pad
    d.dtype, d.split_k = (dtype_code(x) if dtype is None else dtype), split_k
    d.IC_dw, d.OC_dw, d.tile_hint = IC_dw or IC, OC_dw or OC, tile_hint
    d.dw_rot = dw_rot
    if seg2 is not None:
        d.IH2, d.IW2, d.OH2, d.OW2 = seg2
    assert dw.dtype == torch.float32
    return d


def conv_wgrad(x, dy, dw, *, N, IH, IW, IC, OH, OW, OC, KH, KW, stride=1, dil=1, pad=0,
               ld_x=None, ld_dy=None, split_k=0, IC_dw=None, OC_dw=None, tile_hint=0, seg2=None, dtype=None, dw_rot=0):
    d = _wgrad_desc(x, dy, dw, N=N, IH=IH, IW=IW, IC=IC, OH=OH, OW=OW, OC=OC, KH=KH, KW=KW, stride=stride, dil=dil, pad=pad, ld_x=ld_x, ld_dy=ld_dy,
                    split_k=split_k, IC_dw=IC_dw, OC_dw=OC_dw, tile_hint=tile_hint, seg2=seg2, dtype=dtype, dw_rot=dw_rot)
    assert dw.dtype == torch.float32
    if PROFILE_WGRAD is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib.wseg_conv_wgrad(C.byref(d), C.c_void_p(stream_ptr())), "wseg_conv_wgrad")
    if PROFILE_WGRAD is not None:
        ev1.record()
        pix = N * OH * OW + (N * seg2[2] * seg2[3] if seg2 is not None else 0)
        PROFILE_WGRAD.append((ev0, ev1, 2.0 * pix * IC * OC * KH * KW, f"wgrad {IC}->{OC} k{KH} s{stride} d{dil} {OH}x{OW}"))


def pack_weights(master, fwd, tr, OC, T, IC, OCp, ICp, dtype, ic_rot=0):
    check(lib.wseg_pack_weights(C.c_void_p(_ptr(master)), C.c_void_p(_ptr(fwd)), C.c_void_p(_ptr(tr)),
                                OC, T, IC, OCp, ICp, ic_rot, dtype, C.c_void_p(stream_ptr())), "wseg_pack_weights")


def copy2d_batch(src, dst, table, npieces, total_chunks):
    check(lib.wseg_copy2d_batch(C.c_void_p(_ptr(src)), C.c_void_p(_ptr(dst)), C.c_void_p(_ptr(table)), npieces, C.c_long(total_chunks),
                                C.c_void_p(stream_ptr())), "wseg_copy2d_batch")


def pack_transposed_batch(master, out, table, nlayers, total_tiles, dtype):
    check(lib.wseg_pack_transposed_batch(C.c_void_p(_ptr(master)), C.c_void_p(_ptr(out)), C.c_void_p(_ptr(table)), nlayers,
                                         C.c_long(total_tiles), dtype, C.c_void_p(stream_ptr())), "wseg_pack_transposed_batch")


def pack_transposed_batch_bf16(mirror, out, table, nlayers, total_tiles):
    check(lib.wseg_pack_transposed_batch_bf16(C.c_void_p(_ptr(mirror)), C.c_void_p(_ptr(out)), C.c_void_p(_ptr(table)), nlayers,
                                              C.c_long(total_tiles), C.c_void_p(stream_ptr())), "wseg_pack_transposed_batch_bf16")


def dropout_scale(u, out, split_at, p0, p1):
    check(lib.wseg_dropout_scale(C.c_void_p(_ptr(u)), C.c_void_p(_ptr(out)), C.c_long(u.numel()), C.c_long(split_at),
                                 C.c_float(p0), C.c_float(p1), C.c_void_p(stream_ptr())), "wseg_dropout_scale")


def stem_conv_kc(x, w_kc, scale, shift, raw, act, N, H, W, dtype):
    if w_kc.numel() != 27 * 64 or w_kc.dtype != torch.float32:
        raise RuntimeError("stem_conv_kc: weights must be f32 [27][64]")
    check(lib.wseg_stem_conv_kc(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(w_kc)), C.c_void_p(_ptr(scale)), C.c_void_p(_ptr(shift)),
                                C.c_void_p(_ptr(raw)), C.c_void_p(_ptr(act)), N, H, W, dtype,
                                C.c_void_p(stream_ptr())), "wseg_stem_conv_kc")


def stem_conv(x, w, scale, shift, raw, act, N, H, W, dtype):
    check(lib.wseg_stem_conv(C.c_void_p(_ptr(x)), C.c_void_p(_ptr(w)), C.c_void_p(_ptr(scale)), C.c_void_p(_ptr(shift)),
                             C.c_void_p(_ptr(raw)), C.c_void_p(_ptr(act)), N, H, W, dtype,
                             C.c_void_p(stream_ptr())), "wseg_stem_conv")


def _v(x):
    return C.c_void_p(_ptr(x))


def _s():
    return C.c_void_p(stream_ptr())


def head_split(head, ld, c0, cam_low, cmax, N, hw):
    check(lib.wseg_head_split(_v(head), ld, c0, _v(cam_low), _v(cmax), N, hw, dtype_code(head), _s()), "wseg_head_split")


def cam_gate(cam_low, cmax, G, N, hw):
    check(lib.wseg_cam_gate(_v(cam_low), _v(cmax), _v(G), N, hw, _s()), "wseg_cam_gate")


def pcm_xs(x, feat, ld, c_xs, c_end, N, H, W, h, w):
    check(lib.wseg_pcm_xs(_v(x), _v(feat), ld, c_xs, c_end, N, H, W, h, w, dtype_code(feat), _s()), "wseg_pcm_xs")


def head_grad_rows(d_fproj, d_cam_low, head, d_head, ld, N, hw):
    check(lib.wseg_head_grad_rows(_v(d_fproj), _v(d_cam_low), _v(head), _v(d_head), ld, N, hw, dtype_code(head), _s()), "wseg_head_grad_rows")


def resize_planar_fwd(inp, out, planes, ih, iw, oh, ow, align, plane_mul=None, flip_x=False, accumulate=False):
    check(lib.wseg_resize_planar_fwd(_v(inp), _v(out), _v(plane_mul), C.c_long(planes), ih, iw, oh, ow, int(align), int(flip_x), int(accumulate), _s()), "wseg_resize_planar_fwd")


def infer_finish(sum_cam, stats, alpha, norm_cam, pred, npix):
    check(lib.wseg_infer_finish(_v(sum_cam), _v(stats), C.c_float(alpha), _v(norm_cam), _v(pred), npix, _s()), "wseg_infer_finish")


def resize_planar_bwd(d_out, d_in, planes, ih, iw, oh, ow, align, accumulate=False, plane_mul=None, plane_add=None):
    check(lib.wseg_resize_planar_bwd(_v(d_out), _v(d_in), _v(plane_mul), _v(plane_add), C.c_long(planes), ih, iw, oh, ow, int(align), int(accumulate), _s()), "wseg_resize_planar_bwd")


def l2norm_forward(F, ldf, Fh, nrm, rows):
    check(lib.wseg_l2norm_forward(_v(F), ldf, _v(Fh), _v(nrm), C.c_long(rows), dtype_code(F), _s()), "wseg_l2norm_forward")


def l2norm_backward(F, ldf, dFh, nrm, dF, lddf, rows):
    check(lib.wseg_l2norm_backward(_v(F), ldf, _v(dFh), _v(nrm), _v(dF), lddf, C.c_long(rows), dtype_code(F), _s()), "wseg_l2norm_backward")


def pcm_forward(Fh, G, cam_rv, den, N, hw):
    check(lib.wseg_pcm_forward(_v(Fh), _v(G), _v(cam_rv), _v(den), N, hw, _s()), "wseg_pcm_forward")


def pcm_backward(Fh, G, d_cam_rv, cam_rv, den, DN, dFh, N, hw):
    check(lib.wseg_pcm_backward(_v(Fh), _v(G), _v(d_cam_rv), _v(cam_rv), _v(den), _v(DN), _v(dFh), N, hw, _s()), "wseg_pcm_backward")


def sgd_step(params, grads, buf, segs, momentum, grad_scale, first_step, bf16_mirror=None):
    """segs: list of (begin, end, lr, weight_decay) over the flat buffers."""
    n = len(segs)
    LongArr, FloatArr = C.c_long * n, C.c_float * n
    b = LongArr(*[s[0] for s in segs]); e = LongArr(*[s[1] for s in segs])
    lr = FloatArr(*[s[2] for s in segs]); wd = FloatArr(*[s[3] for s in segs])
    check(lib.wseg_sgd_step(_v(params), _v(grads), _v(buf), C.c_long(params.numel()), b, e, lr, wd, n,
                            C.c_float(momentum), C.c_float(grad_scale), int(first_step), _v(bf16_mirror), _s()), "wseg_sgd_step")


# ---------------------------------------------------------------------------------------------- loss kernels
lib.wseg_select_workspace_bytes.restype = C.c_size_t


def _call(name, *args):
    check(getattr(lib, name)(*args, _s()), name)


def _f(x):
    return C.c_float(x)


lib.wseg_plane_stats_workspace_bytes.restype = C.c_size_t


def plane_stats(U, stats, planes, npix):
    ws = torch.empty(int(lib.wseg_plane_stats_workspace_bytes(C.c_long(planes))), device=U.device, dtype=torch.uint8)
    _call("wseg_plane_stats", _v(U), _v(stats), C.c_long(planes), npix, _v(ws))
def cls_loss(stats, label20, loss_out, plane_bias, N, npix, coef): _call("wseg_cls_loss", _v(stats), _v(label20), _v(loss_out), _v(plane_bias), N, npix, _f(coef))
def rvmin_values(U, label20, q, argc, N, npix): _call("wseg_rvmin_values", _v(U), _v(label20), _v(q), _v(argc), N, npix)
def select_workspace_bytes(rows): return int(lib.wseg_select_workspace_bytes(rows))
def select_kth(vals, rows, n, k, largest, use_abs, relu_vals, res, ws): _call("wseg_select_kth", _v(vals), rows, n, k, int(largest), int(use_abs), int(relu_vals), _v(res), _v(ws))
def loss_finish(acc, er_coef, out8): _call("wseg_loss_finish", _v(acc), _f(er_coef), _v(out8))
def select_finish(res, rows, k, relu_vals, scale, loss_out): _call("wseg_select_finish", _v(res), rows, k, int(relu_vals), _f(scale), _v(loss_out))
def rvmin_backward(q, argc, res, label20, dU, N, npix, k, coef): _call("wseg_rvmin_backward", _v(q), _v(argc), _v(res), _v(label20), _v(dU), N, npix, k, _f(coef))
def norm_resize_forward(U, stats, label20, out, N, S, OS): _call("wseg_norm_resize_forward", _v(U), _v(stats), _v(label20), _v(out), N, S, OS)
def norm_resize_backward(G, U, stats, label20, dU, N, S, OS): _call("wseg_norm_resize_backward", _v(G), _v(U), _v(stats), _v(label20), _v(dU), N, S, OS)
def er_ecr_prep(c1, c2, r1, r2, Gc1, Gc2, dlt1, dlt2, er_sum, N, npix, er_coef): _call("wseg_er_ecr_prep", _v(c1), _v(c2), _v(r1), _v(r2), _v(Gc1), _v(Gc2), _v(dlt1), _v(dlt2), _v(er_sum), N, npix, _f(er_coef))
def ecr_backward(dlt, res, Gr, N, per_row, k, coef): _call("wseg_ecr_backward", _v(dlt), _v(res), _v(Gr), N, per_row, k, _f(coef))
def rows_resize_forward(head, ld, F, N, ih, iw, oh, ow): _call("wseg_rows_resize_forward", _v(head), ld, _v(F), N, ih, iw, oh, ow, dtype_code(head))
def head_grad_fused(dF, d_cam_low, head, d_head, ld, N, ih, iw, oh, ow): _call("wseg_head_grad_fused", _v(dF), _v(d_cam_low), _v(head), _v(d_head), ld, N, ih, iw, oh, ow, dtype_code(head))
def pseudo_label(R, label20, bg_thr, y, ncam, N, npix): _call("wseg_pseudo_label", _v(R), _v(label20), _f(bg_thr), _v(y), _v(ncam), N, npix)
def proto_candidates(ncam, F, tie_idx, cand_val, cand_feat, cand_const, N, npix, K): _call("wseg_proto_candidates", _v(ncam), _v(F), _v(tie_idx), _v(cand_val), _v(cand_feat), _v(cand_const), N, npix, K)
def proto_merge(cand_val, cand_feat, cand_const, protos, world, K, rank_stride=0): _call("wseg_proto_merge", _v(cand_val), _v(cand_feat), _v(cand_const), _v(protos), world, K, C.c_long(rank_stride))
def nce_sims(F, p_own, p_oth, fn, nrm, S_own, S_oth, P): _call("wseg_nce_sims", _v(F), _v(p_own), _v(p_oth), _v(fn), _v(nrm), _v(S_own), _v(S_oth), P)
def intra_weights(y, S_own, rkey, rand_flag, w, P, ld_s=21): _call("wseg_intra_weights", _v(y), _v(S_own), ld_s, _v(rkey), _v(rand_flag), _v(w), P)
def intra_pack(y, S_own, rkey, rec, P): _call("wseg_intra_pack", _v(y), _v(S_own), _v(rkey), _v(rec), P)
def intra_weights_global(rec, w, P, ranks, own_rank, scale, rank_stride): _call("wseg_intra_weights_global", _v(rec), _v(w), P, ranks, own_rank, _f(scale), C.c_long(rank_stride))
def nce_loss_grad(fn, nrm, S_own, S_oth, y_own, y_oth, w_intra, p_own, p_oth, dF, sums, P, coef_cross, coef_intra):
    _call("wseg_nce_loss_grad", _v(fn), _v(nrm), _v(S_own), _v(S_oth), _v(y_own), _v(y_oth), _v(w_intra), _v(p_own), _v(p_oth), _v(dF), _v(sums), P, _f(coef_cross), _f(coef_intra))


class NceView(C.Structure):
    _fields_ = [("F", C.c_void_p), ("p_own", C.c_void_p), ("p_oth", C.c_void_p), ("y_own", C.c_void_p), ("y_oth", C.c_void_p),
                ("w_intra", C.c_void_p), ("rkey", C.c_void_p), ("rec", C.c_void_p), ("dF", C.c_void_p)]


def _nce_views(views):
    arr = (NceView * len(views))()
    for i, v in enumerate(views):
        for k, _t in NceView._fields_:
            setattr(arr[i], k, _ptr(v.get(k)))
    return arr


def nce_records(views, P, split_bf16=False):
    """views: list of dicts (F, p_own, y_own, rec[, rkey]) — one launch for all of them."""
    _call("wseg_nce_records", _nce_views(views), len(views), P, int(split_bf16))


def nce_fused(views, P, coef_cross, coef_intra, sums):
    """views: list of dicts (F, p_own, p_oth, y_own, y_oth, w_intra, dF) — one launch for all of them."""
    _call("wseg_nce_fused", _nce_views(views), len(views), P, _f(coef_cross), _f(coef_intra), _v(sums))


# ---- SEAM map losses evaluated on the fly from the stride-8 maps (csrc/maps.hip): no [N,21,S,S] tensors
def up_plane_stats(low, stats, planes, h, w, S, label20=None):
    ws = torch.empty(int(lib.wseg_plane_stats_workspace_bytes(C.c_long(planes))), device=low.device, dtype=torch.uint8)
    _call("wseg_up_plane_stats", _v(low), _v(stats), C.c_long(planes), h, w, S, _v(label20), _v(ws))
def up_rvmin_values(low, label20, q, argc, N, h, w, S): _call("wseg_up_rvmin_values", _v(low), _v(label20), _v(q), _v(argc), N, h, w, S)
def up_norm_resize_forward(low, stats, label20, out, N, h, w, S, OS): _call("wseg_up_norm_resize_forward", _v(low), _v(stats), _v(label20), _v(out), N, h, w, S, OS)
def resize_adjoint_ones(wvec, h, S): _call("wseg_resize_adjoint_ones", _v(wvec), h, S)
def up_maps_backward(G, low, stats, label20, plane_bias, wvec_y, wvec_x, q, argc, res, k, coef, d_low, N, h, w, S, OS):
    _call("wseg_up_maps_backward", _v(G), _v(low), _v(stats), _v(label20), _v(plane_bias), _v(wvec_y), _v(wvec_x), _v(q), _v(argc), _v(res),
          k, _f(coef), _v(d_low), N, h, w, S, OS)


def gemm256_probe(A, B, Cout, M, N, K, variant=0):
    if not hasattr(lib, "wseg_gemm256_probe"):
        raise RuntimeError("wseg_gemm256_probe is a development probe: rebuild with `WSEG_PROBES=1 bash wseg_amd/csrc/build.sh`")
    check(lib.wseg_gemm256_probe(_v(A), _v(B), _v(Cout), M, N, K, variant, _s()), "wseg_gemm256_probe")


def debug_stamps(n_wg):
    """probe builds: the in-kernel stamps of the last 256-tile conv launch, [n_wg, 24] uint64: 8 wall-clock stamps (100 MHz) + 2 x 8 slot cycle sums in WSEG_PROBES=2 builds (csrc/conv_igemm.hip)"""
    if not hasattr(lib, "wseg_debug_stamps"):
        raise RuntimeError("wseg_debug_stamps exists in probe builds only: rebuild with `WSEG_PROBES=1 bash wseg_amd/csrc/build.sh`")
    import numpy as np
    torch.cuda.synchronize()
    buf = np.zeros((min(n_wg, 4096), 24), dtype=np.uint64)
    lib.wseg_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
    check(lib.wseg_debug_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)), "wseg_debug_stamps")
    return buf


def debug_set_diag(v):
    """probe builds: timing switches of the 256-tile conv main loop (results wrong by design; csrc/conv_igemm.hip g_wseg_diag)"""
    if not hasattr(lib, "wseg_debug_set_diag"):
        raise RuntimeError("wseg_debug_set_diag exists in probe builds only")
    check(lib.wseg_debug_set_diag(int(v)), "wseg_debug_set_diag")


def to_bf16(inp, out): _call("wseg_to_bf16", _v(inp), _v(out), C.c_long(inp.numel()))
def split_bf16(inp, hi, lo): _call("wseg_split_bf16", _v(inp), _v(hi), _v(lo), C.c_long(inp.numel()))
def pack_x3(src, dst): _call("wseg_pack_x3", _v(src), _v(dst), C.c_long(src.numel()))
def pcm_forward_bf16(Fb, Gb, cam_rv, den, N, hw): _call("wseg_pcm_forward_bf16", _v(Fb), _v(Gb), _v(cam_rv), _v(den), N, hw)
def pcm_backward_bf16(Fb, Gb, Gl, d_cam_rv, cam_rv, den, DN, DNb, DNl, dFh, N, hw): _call("wseg_pcm_backward_bf16", _v(Fb), _v(Gb), _v(Gl), _v(d_cam_rv), _v(cam_rv), _v(den), _v(DN), _v(DNb), _v(DNl), _v(dFh), N, hw)
