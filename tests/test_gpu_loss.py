"""GPU parity of the full training step (Net fwd/bwd on HIP + loss) against fixtures produced by the
REFERENCE ITSELF (tests/golden/step_*.npz, see oracle/make_goldens.py): the 8 logged scalars within
1e-4 (fp32 mode, north-star tolerance) and weight-gradient slices, for both loss back-ends:
  hip  — the hand-written loss kernels (csrc/loss.hip), the product path;
  aten — the cross-check path (tests/aten_trainer.py: device-side torch ops behind the drop-in Net.forward 4-tuple)."""
import contextlib
import io
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SCALARS = ["loss", "loss_cls", "loss_er", "loss_ecr", "loss_nce", "loss_intra_nce", "loss_cross_nce", "loss_cross_nce2"]


def _trainer(proc_sd, precision, loss_impl, n, seed, py_seed, lr=0.01):
    from wseg_amd import synth
    from wseg_amd.loss_hip import cpu_tie_pattern
    from wseg_amd.optim import PolyOptimizer
    from wseg_amd.resnet38_contrast import Net
    from wseg_amd.train import Trainer
    from .aten_trainer import AtenTrainer
    model = Net(precision=precision)
    with contextlib.redirect_stdout(io.StringIO()):
        groups = model.get_parameter_groups()
    opt = PolyOptimizer([{'params': groups[0], 'lr': lr, 'weight_decay': 5e-4}, {'params': groups[1], 'lr': 2 * lr, 'weight_decay': 0},
                         {'params': groups[2], 'lr': 10 * lr, 'weight_decay': 5e-4}, {'params': groups[3], 'lr': 20 * lr, 'weight_decay': 0}],
                        lr=lr, weight_decay=5e-4, max_step=100)
    model.load_state_dict(proc_sd)
    model.cuda()
    model.train()
    model.set_dropout_masks([synth.synthetic_dropout_masks(n, seed * 2 + 0), synth.synthetic_dropout_masks(n, seed * 2 + 1)])
    cls = Trainer if loss_impl == "hip" else AtenTrainer
    tr = cls(model, opt, 0.20, random.Random(py_seed), rng_parity=True, bg_topk_idx=cpu_tie_pattern(n * 256, 32))
    return model, opt, tr


@pytest.mark.parametrize("loss_impl,prec", [("hip", "fp32"), ("aten", "fp32"), ("hip", "bf16x3")])
@pytest.mark.parametrize("name", ["step_S160_N2", "step_S128_N3", "step_edge_S64_N3"])
def test_step_matches_reference_fixture(golden_dir, proc_sd, name, loss_impl, prec):
    """prec bf16x3: the SAME bars (north-star tolerance 1e-4 on the scalars) with every conv / weight-gradient product computed as
    split-bf16 (3 bf16 MFMAs, 16-17 operand bits) instead of the exact-f32 MFMA."""
    from wseg_amd import synth
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
    model, opt, tr = _trainer(proc_sd, prec, loss_impl, n, seed, py_seed)
    w_before = model._engine.conv_param("fc8").detach().clone() if model._engine.flat_w is not None else None
    lab = torch.from_numpy(g["label"]) if "label" in g.files else synth.synthetic_labels(n, seed)   # (edge fixture: an image with no
    got = tr.step(synth.synthetic_images(n, size, seed).cuda(), lab.cuda())                            #  class, one with all twenty)
    for k in SCALARS:
        ref = float(g["s/" + k])
        assert abs(float(got[k]) - ref) <= 1e-4 * max(1.0, abs(ref)), (k, float(got[k]), ref)
    params = dict(model.named_parameters())
    # Head-layer weight gradients depend on the forward values and the loss gradient only: 2e-3 of the slice maximum for every
    # fixture.  Backbone weight gradients also pass through the ReLU masks of every later layer; the edge fixture's
    # all-twenty-classes image at 64x64 puts dozens of pre-activations within f32 summation noise of zero under 10-50 % of
    # their layer's largest gradient (scripts/relu_near_ties.py), and which of them flip depends on the summation order, so
    # its backbone bar is 5e-2 — a mishandled empty / full label row would show in the scalars and the head gradients.
    head = ("fc8.", "fc_proj.", "f9.", "f8_3.", "f8_4.")
    # split-bf16 mode: its forward differs from the reference's by ~1e-5 instead of ~1e-7, so MORE ReLU pre-activations near zero resolve
    # differently than in the reference (the effect the edge fixture shows in fp32) — the arithmetic itself is held to 2e-3 by
    # test_gradients_under_the_hip_paths_relu_decisions[bf16x3]; measured worst slice error 1.1e-2 (profiles/r02_bf16x3_deviation.json)
    wide = 5e-2 if "edge" in name else (2.5e-2 if prec == "bf16x3" else 2e-3)
    for key in g.files:
        if not key.startswith("gslice/"):
            continue
        k = key[len("gslice/"):]
        tol = 2e-3 if k.startswith(head) else wide
        gr = params[k].grad.detach().cpu()
        flat = gr.reshape(-1)
        stepv = max(1, flat.numel() // 4096)
        ref = g[key]
        scale = np.abs(ref).max() + 1e-12
        assert np.abs(flat[::stepv][:4096].numpy() - ref).max() / scale < tol, k
        gn = float(g["gnorm/" + k])
        assert abs(float(gr.double().norm()) - gn) < tol * gn, k
    assert opt.global_step == 1


GRAD_KEYS = ["fc8.weight", "fc_proj.weight", "f9.weight", "f8_3.weight", "f8_4.weight", "b7.conv_branch2b1.weight",
             "b7.conv_branch1.weight", "b6.conv_branch2a.weight", "b5.conv_branch2a.weight", "b4.conv_branch2a.weight",
             "b4.conv_branch1.weight", "b3.conv_branch2a.weight", "b3_1.conv_branch2b1.weight"]


def gates_from_ctx(S):
    """The ReLU decisions the HIP forward took, per view, in the oracle's layout (0/1 float NCHW tensors keyed as oracle/net.py
    `_relu` keys them) — read off the saved activations of the engine's forward context."""
    from wseg_amd import arch
    N, V = S["N"], S["V"]
    out = [dict() for _ in range(V)]

    def put(key, rows, dims, cols=None):
        off = 0
        for vi, (h, w) in enumerate(dims):
            r = rows[off:off + N * h * w]
            off += N * h * w
            if cols is not None:
                r = r[:, cols[0]:cols[1]]
            out[vi][key] = (r.float() != 0).view(N, h, w, -1).permute(0, 3, 1, 2).float().cpu()

    for b in arch.BLOCKS:
        name = b[0]
        din, dout = S["dims"][name]
        for k, dims in (("t", din), ("v", dout), ("v1", dout), ("v2", dout)):
            if k in S[name]:
                put(f"{name}.{k}", S[name][k], dims)
    put("conv6", S["fea"], S["hdims"])
    put("f_proj", S["head"], S["hdims"], (0, 128))
    put("f8_3", S["feat"], S["hdims"], (0, 64))
    put("f8_4", S["feat"], S["hdims"], (64, 192))
    return out


@pytest.mark.parametrize("name,prec", [("step_edge_S64_N3", "fp32"), ("step_S160_N2", "bf16x3"), ("step_edge_S64_N3", "bf16x3")])
def test_gradients_under_the_hip_paths_relu_decisions(golden_dir, proc_sd, name, prec):
    """(also run in the split-bf16 mode, whose ~1e-5 forward deviation flips more near-zero pre-activations than exact f32 does.)
    Why the edge fixture's backbone gradients are held to 5e-2 against the reference and not 2e-3: its all-twenty-classes
    image puts ReLU pre-activations within f32 summation noise of zero under large gradients, and which of them pass depends on
    the summation order (scripts/relu_near_ties.py).  Demonstrated here rather than argued: the CPU oracle re-run with the HIP
    path's OWN ReLU decisions injected (every backbone / head ReLU site, both views) must reproduce the HIP gradients at the
    ordinary 2e-3 bar for all 13 fixture keys — so the only difference between the two implementations is which near-zero
    pre-activations they let through, not the arithmetic."""
    from oracle import loss as oloss
    from oracle import net as onet
    from wseg_amd import synth
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
    model, opt, tr = _trainer(proc_sd, prec, "hip", n, seed, py_seed)
    model._engine.capture_ctx = True
    lab = torch.from_numpy(g["label"]) if "label" in g.files else synth.synthetic_labels(n, seed)
    img = synth.synthetic_images(n, size, seed)
    got = tr.step(img.cuda(), lab.cuda())
    gates = gates_from_ctx(model._engine.last_ctx)
    model._engine.capture_ctx, model._engine.last_ctx = False, None
    sd = {k: v.clone() for k, v in proc_sd.items()}
    for k in onet.trainable_keys(sd):
        sd[k].requires_grad_(True)
    ref = oloss.train_step(img, lab, sd, synth.synthetic_dropout_masks(n, seed * 2), synth.synthetic_dropout_masks(n, seed * 2 + 1),
                           0.20, random.Random(py_seed), gates1=gates[0], gates2=gates[1])
    ref["loss"].backward()
    for k in SCALARS:
        assert abs(float(got[k]) - float(ref[k])) <= 1e-4 * max(1.0, abs(float(ref[k]))), (k, float(got[k]), float(ref[k]))
    params = dict(model.named_parameters())
    flipped = 0
    for k in GRAD_KEYS:
        a = params[k].grad.detach().cpu().reshape(-1)
        b = sd[k].grad.reshape(-1)
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) / scale < 2e-3, (k, float((a - b).abs().max()) / scale)
        assert abs(float(a.double().norm()) - float(b.double().norm())) < 2e-3 * float(b.double().norm()), k
        fix = g["gslice/" + k]
        stepv = max(1, a.numel() // 4096)
        flipped += int(np.abs(a[::stepv][:4096].numpy() - fix).max() / (np.abs(fix).max() + 1e-12) > 2e-3)
    print(f"{name} [{prec}]: {flipped} of {len(GRAD_KEYS)} keys differ from the reference fixture by more than 2e-3 (none from the gate-injected oracle)")


# Full resolution (BASELINE config 2: 448 x 448, 56 x 56 maps), two draws, fixtures from the reference.  At this resolution the per-class top-32 of
# 512 contrast pixels (contrast_train.py:202-203) regularly holds a near-tie at its boundary: in both fixtures the smallest relative gap between the
# 32nd and the 33rd value of some class is 1.6e-5, inside the ~1e-5 by which the PCM-refined CAM of two exact-f32 implementations differs (the CAM
# gate of resnet38_contrast.py:46-48 is discontinuous).  One membership swap moves that class's prototype by ~4e-3 and the NCE terms by ~1e-4 at
# P = 512 — measured on step_S448_N2: every pseudo-label equal, 41 of 42 prototypes equal to 6e-6, class 15 of view 1 off by 4.4e-3, loss_nce off by
# 1.12e-4 (scripts/diag_s448.py).  So: (a) against the reference fixture the non-NCE scalars are held to 1e-4 and the NCE terms to 4e-4;
# (b) the arithmetic is held to 1e-4 on ALL 8 scalars by the selection-injected oracle: the CPU oracle re-run with the HIP path's own prototypes
# and pseudo-labels in place of its own.
S448_NCE_BAR = 4e-4
REF_TIE = {"fp32": 1e-4, "bf16x3": 1e-4}     # largest REFERENCE gap at which a selection may differ; measured (profiles/r03_parity_measured.txt): 1.4e-5 in both modes


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name", ["step_S448_N2", "step_S448_N2_b"])
def test_full_resolution_step_against_reference_fixture(golden_dir, proc_sd, name, prec):
    from oracle import loss as oloss
    from wseg_amd import synth
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
    model, opt, tr = _trainer(proc_sd, prec, "hip", n, seed, py_seed)
    model._engine.capture_ctx = True
    img, lab = synth.synthetic_images(n, size, seed), synth.synthetic_labels(n, seed)
    got = tr.step(img.cuda(), lab.cuda())
    v1, v2 = model._engine.last_loss_views
    model._engine.capture_ctx, model._engine.last_ctx, model._engine.last_loss_views = False, None, None
    nce = ("loss", "loss_nce", "loss_intra_nce", "loss_cross_nce", "loss_cross_nce2")
    for k in SCALARS:
        ref = float(g["s/" + k])
        bar = S448_NCE_BAR if k in nce else 1e-4
        assert abs(float(got[k]) - ref) <= bar * max(1.0, abs(ref)), (k, float(got[k]), ref)
    for v, key in ((v1, "pseudo1"), (v2, "pseudo2")):
        assert float((v.y.cpu().numpy() != g[key].astype(np.int32)).mean()) <= 1 / 256, key            # (pseudo-labels: at most a near-tie pixel or two)
    swaps = sum(int((np.abs(v.protos.cpu().numpy() - g[key]).max(axis=1) > 1e-4).sum()) for v, key in ((v1, "protos1"), (v2, "protos2")))
    assert swaps <= 3, swaps
    # ... and every difference is a near-tie OF THE REFERENCE: tests/golden/<name>_margins.npz (oracle/make_goldens.py step_margins_golden) holds the reference's own
    # 32nd / 33rd value of each class's top-k input and its own top-1 / top-2 gap of every pseudo-label; a class whose prototype differs, or a pixel whose label
    # differs, must sit on a reference gap below REF_TIE — a statement about the reference's numbers, not a bound fitted to this implementation's
    mg = np.load(os.path.join(golden_dir, name + "_margins.npz"))
    worst_gap = 0.0
    for vi, (v, pk, yk) in enumerate(((v1, "protos1", "pseudo1"), (v2, "protos2", "pseudo2")), start=1):
        t33 = mg["top33_%d" % vi].astype(np.float64)
        for c in np.nonzero(np.abs(v.protos.cpu().numpy() - g[pk]).max(axis=1) > 1e-4)[0]:
            gap = float(t33[c, 31] - t33[c, 32])
            worst_gap = max(worst_gap, gap)
            assert gap <= REF_TIE[prec], ("prototype of class", int(c), "view", vi, "reference 32nd/33rd gap", gap)
        for px in np.nonzero(v.y.cpu().numpy() != g[yk].astype(np.int32))[0]:
            gap = float(mg["label_margin_%d" % vi][px])
            worst_gap = max(worst_gap, gap)
            assert gap <= REF_TIE[prec], ("pseudo-label of pixel", int(px), "view", vi, "reference top-1/top-2 gap", gap)
    inject = dict(protos1=v1.protos.cpu(), protos2=v2.protos.cpu(), pseudo1=v1.y.cpu().long(), pseudo2=v2.y.cpu().long())
    with torch.no_grad():
        ref = oloss.train_step(img, lab, dict(proc_sd), synth.synthetic_dropout_masks(n, seed * 2), synth.synthetic_dropout_masks(n, seed * 2 + 1),
                               0.20, random.Random(py_seed), inject=inject)
    for k in SCALARS:
        assert abs(float(got[k]) - float(ref[k])) <= 1e-4 * max(1.0, abs(float(ref[k]))), ("selection-injected", k, float(got[k]), float(ref[k]))
    print(f"{name} [{prec}]: {swaps} prototype(s) differ from the reference by a top-32 near-tie (largest reference gap among the differing selections {worst_gap:.2e}); "
          f"all 8 scalars within 1e-4 of the selection-injected oracle")


def test_bf16_full_resolution_step(golden_dir, proc_sd):
    """Throughput mode at the real resolution against the reference's fixtures: measured relative deviations (profiles/r02_bf16_deviation.json) loss 1.7e-2,
    cls 1e-4, er 1e-3, ecr 1.4e-3, nce 4.4e-2, intra 8.5e-2, cross 1.7e-3, cross2 6.5e-2 — the NCE terms move through prototype / pseudo-label
    selections (discrete), which bf16 forward noise changes; bars = 2x measured."""
    from wseg_amd import synth
    bars = {"loss": 3.4e-2, "loss_cls": 4.2e-3, "loss_er": 1.5e-2, "loss_ecr": 2.7e-2, "loss_nce": 9e-2, "loss_intra_nce": 1.7e-1,
            "loss_cross_nce": 2.3e-2, "loss_cross_nce2": 1.3e-1}
    for name in ("step_S448_N2", "step_S448_N2_b"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
        model, opt, tr = _trainer(proc_sd, "bf16", "hip", n, seed, py_seed)
        got = tr.step(synth.synthetic_images(n, size, seed).cuda(), synth.synthetic_labels(n, seed).cuda())
        for k in SCALARS:
            ref = float(g["s/" + k])
            assert abs(float(got[k]) - ref) <= bars[k] * abs(ref), (name, k, float(got[k]), ref)


# ---- the benchmarked mode (bf16) against what THE REFERENCE ITSELF shows in bf16 ------------------------------------------------------------
# tests/golden/<name>_refbf16.npz (oracle/make_goldens.py `step_refbf16_golden`): the reference Net run with `model.bfloat16()` (weights and
# activations bf16, loss maths f32) on the fixture's inputs.  Its deviation from its own fp32 run is the envelope of the storage precision: discrete
# selections (CAM gate, pseudo-labels, top-32 members, ReLU decisions) flip under bf16 rounding in ANY implementation.  Per scalar, the envelope is
# the reference's worst deviation over the three fixtures; the HIP bf16 mode must stay within ENVELOPE_FACTOR x of it on every fixture.  Per gradient
# key group: cosine / norm-ratio DEFECT (1 - cos, |ratio - 1|) within the same factor of the reference's worst defect in that group.
ENVELOPE_FIXTURES = ("step_S160_N2", "step_S448_N2", "step_S448_N2_b")
ENVELOPE_FACTOR = 1.5
# the four NCE terms (and the total, which contains them) hang on discrete selections over the 512 contrast pixels of a 2-image fixture — pseudo-labels,
# top-32 members, hard-pixel rank bands — whose flips are a small-sample draw in BOTH implementations: measured worst ratio 1.57 (loss_intra_nce of
# step_S448_N2: 8.5e-2 against the reference's own 5.4e-2); every other quantity sits inside 1.5 x
ENVELOPE_FACTOR_NCE = 2.0
# the reference's bf16 run overflows fc8's gradient at 448 x 448, so the fc8 group's envelope comes from the 160 x 160 fixture alone (1 - cos = 1.3e-5); a cosine
# defect below 1e-4 (cos >= 0.9999) is bf16 rounding of the operands at any size (measured on step_S448_N2: 3.3e-5) and is accepted as such
ENVELOPE_COS_FLOOR = 1e-4
NCE_SCALARS = ("loss", "loss_nce", "loss_intra_nce", "loss_cross_nce", "loss_cross_nce2")


def _grad_group(key):
    if key.startswith(("f9.", "f8_3.", "f8_4.")):
        return "pcm"
    for k in ("fc8.", "fc_proj."):
        if key.startswith(k):
            return k
    return "backbone"


def _dev_of(scal, grads, gnorms, g):
    """(per-scalar relative deviation, per-key (1 - cosine, |norm ratio - 1|)) of a run against the fp32 fixture g"""
    ds = {k: abs(scal[k] - float(g["s/" + k])) / abs(float(g["s/" + k])) for k in SCALARS}
    dg = {}
    for k in GRAD_KEYS:
        a, b = grads[k].astype(np.float64), g["gslice/" + k].astype(np.float64)
        if not (np.isfinite(a).all() and np.isfinite(gnorms[k])):
            continue                                     # (the reference's bf16 run overflows fc8's gradient at 448 x 448: no envelope from that key)
        dg[k] = (1.0 - float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300)), abs(gnorms[k] / float(g["gnorm/" + k]) - 1.0))
    return ds, dg


def test_bf16_within_the_reference_bf16_envelope(golden_dir, proc_sd):
    from wseg_amd import synth
    env_s = {k: 0.0 for k in SCALARS}
    env_g = {}
    ours = {}
    for name in ENVELOPE_FIXTURES:
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        rb = np.load(os.path.join(golden_dir, name + "_refbf16.npz"))
        ds, dg = _dev_of({k: float(rb["s/" + k]) for k in SCALARS}, {k: rb["gslice/" + k] for k in GRAD_KEYS},
                         {k: float(rb["gnorm/" + k]) for k in GRAD_KEYS}, g)
        for k in SCALARS:
            env_s[k] = max(env_s[k], ds[k])
        for k, (c, r) in dg.items():
            e = env_g.setdefault(_grad_group(k), [0.0, 0.0])
            e[0], e[1] = max(e[0], c), max(e[1], r)
        n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
        model, opt, tr = _trainer(proc_sd, "bf16", "hip", n, seed, py_seed)
        got = tr.step(synth.synthetic_images(n, size, seed).cuda(), synth.synthetic_labels(n, seed).cuda())
        params = dict(model.named_parameters())
        gs, gn = {}, {}
        for k in GRAD_KEYS:
            flat = params[k].grad.detach().cpu().reshape(-1)
            gs[k] = flat[::max(1, flat.numel() // 4096)][:4096].numpy()
            gn[k] = float(params[k].grad.double().norm())
        ours[name] = _dev_of({k: float(got[k]) for k in SCALARS}, gs, gn, g)
    worst, bad = {}, []
    for name, (ds, dg) in ours.items():
        for k in SCALARS:
            worst[k] = max(worst.get(k, 0.0), ds[k] / env_s[k])
            if ds[k] > (ENVELOPE_FACTOR_NCE if k in NCE_SCALARS else ENVELOPE_FACTOR) * env_s[k]:
                bad.append((name, k, ds[k], env_s[k]))
        for k, (c, r) in dg.items():
            ec, er = env_g[_grad_group(k)]
            worst["cos:" + _grad_group(k)] = max(worst.get("cos:" + _grad_group(k), 0.0), c / ec)
            worst["norm:" + _grad_group(k)] = max(worst.get("norm:" + _grad_group(k), 0.0), r / er)
            if c > ENVELOPE_FACTOR * ec + ENVELOPE_COS_FLOOR:
                bad.append((name, k, "1 - cos", c, ec))
            if r > ENVELOPE_FACTOR * er + 2e-3:
                bad.append((name, k, "|norm ratio - 1|", r, er))
    print("HIP bf16 deviation / reference-bf16 envelope (worst over fixtures):", {k: round(v, 3) for k, v in worst.items()})
    print("envelope:", {k: float("%.3g" % v) for k, v in env_s.items()}, {k: [float("%.3g" % x) for x in v] for k, v in env_g.items()})
    assert not bad, bad


# ... and the ARITHMETIC of the bf16 mode held tight: the CPU oracle (f32) re-run with the bf16 path's own discrete decisions — every ReLU site, the
# gated CAM that enters the PCM, pseudo-labels and prototypes — must agree with the bf16 step to bf16-rounding size; what is left of the distance to the
# reference fixture above is then flips of those decisions, not arithmetic (a bug in e.g. the bf16 PCM backward would show here, not hide in a 0.5 bar).
BF16_INJECTED_SCALAR_BAR = 1.0e-2      # relative; measured worst 3.1e-3 (loss_ecr) — see the test's printed line
# PCM branch (f9 / f8_3 / f8_4): measured 0.911-0.951 (round 3, scripts/diag_f9_bf16.py: uniform over f9's column groups and rows, i.e. rounding noise of the bf16
# affinity products, not a layout error; 1.0000 in bf16x3) — the reference's own bf16 run has 0.32-0.54 against its fp32 run on these keys (tests/golden/*_refbf16.npz)
BF16_INJECTED_COS_BAR = {"fc8.": 0.9995, "fc_proj.": 0.995, "pcm": 0.88, "backbone": 0.985}
BF16_INJECTED_NORM_BAR = 0.03
BF16_INJECTED_NORM_BAR_PCM = 0.05      # measured worst 0.034 (f8_4 of step_S160_N2)


@pytest.mark.parametrize("name", ["step_S160_N2", "step_S128_N3"])
def test_bf16_arithmetic_under_its_own_decisions(golden_dir, proc_sd, name):
    from oracle import loss as oloss
    from oracle import net as onet
    from wseg_amd import synth
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
    model, opt, tr = _trainer(proc_sd, "bf16", "hip", n, seed, py_seed)
    eng = model._engine
    eng.capture_ctx = True
    img, lab = synth.synthetic_images(n, size, seed), synth.synthetic_labels(n, seed)
    got = tr.step(img.cuda(), lab.cuda())
    S = eng.last_ctx
    gates = gates_from_ctx(S)
    for vi, vw in enumerate(S["views"]):                 # the gated, normalised CAM the HIP forward fed its PCM (rows [pixels][32], 21 used)
        h, w, off = vw["h"], vw["w"], vw["off"]
        gates[vi]["cam_d_norm"] = S["G"][off:off + n * h * w, :21].float().view(n, h, w, 21).permute(0, 3, 1, 2).contiguous().cpu()
    v1, v2 = eng.last_loss_views
    inject = dict(protos1=v1.protos.cpu(), protos2=v2.protos.cpu(), pseudo1=v1.y.cpu().long(), pseudo2=v2.y.cpu().long())
    eng.capture_ctx, eng.last_ctx, eng.last_loss_views = False, None, None
    sd = {k: v.clone() for k, v in proc_sd.items()}
    for k in onet.trainable_keys(sd):
        sd[k].requires_grad_(True)
    ref = oloss.train_step(img, lab, sd, synth.synthetic_dropout_masks(n, seed * 2), synth.synthetic_dropout_masks(n, seed * 2 + 1),
                           0.20, random.Random(py_seed), gates1=gates[0], gates2=gates[1], inject=inject)
    ref["loss"].backward()
    meas, bad = {}, []
    for k in SCALARS:
        meas[k] = abs(float(got[k]) - float(ref[k])) / abs(float(ref[k]))
        if meas[k] > BF16_INJECTED_SCALAR_BAR:
            bad.append((k, float(got[k]), float(ref[k])))
    params = dict(model.named_parameters())
    for k in GRAD_KEYS:
        a = params[k].grad.detach().cpu().reshape(-1).double()
        b = sd[k].grad.reshape(-1).double()
        cos = float(a @ b / (a.norm() * b.norm() + 1e-300))
        ratio = float(a.norm() / b.norm())
        meas["cos:" + k] = cos
        meas["norm:" + k] = ratio
        if cos < BF16_INJECTED_COS_BAR[_grad_group(k)]:
            bad.append((k, "cos", cos))
        if abs(ratio - 1.0) > (BF16_INJECTED_NORM_BAR_PCM if _grad_group(k) == "pcm" else BF16_INJECTED_NORM_BAR):
            bad.append((k, "norm ratio", ratio))
    print(f"{name} [bf16 vs the oracle under its own decisions]:", {k: float("%.4g" % v) for k, v in meas.items()})
    assert not bad, bad


def _multistep(proc_sd, g, prec, loss_impl="hip"):
    """The 3-step fixture's protocol on the HIP path; returns ([per-step scalar dict], {key: relative error of the weight
    DELTA w_after - w_before on the fixture's 4096-sample slice})."""
    from wseg_amd import synth
    n, size, seed, py_seed, steps = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"]), int(g["steps"])
    model, opt, tr = _trainer(proc_sd, prec, loss_impl, n, seed, py_seed, lr=float(g["lr"]))
    opt.max_step = int(g["max_step"])
    masks = []
    for s_ in range(steps):
        masks += [synth.synthetic_dropout_masks(n, (seed + s_) * 2), synth.synthetic_dropout_masks(n, (seed + s_) * 2 + 1)]
    model.set_dropout_masks(masks)
    scal = []
    for s_ in range(steps):
        got = tr.step(synth.synthetic_images(n, size, seed + s_).cuda(), synth.synthetic_labels(n, seed + s_).cuda())
        scal.append({k: float(got[k]) for k in SCALARS})
    assert opt.global_step == steps
    np.testing.assert_allclose([gr["lr"] for gr in opt.param_groups], g["lr_final"], rtol=1e-12)
    params = dict(model.named_parameters())
    dw = {}
    for key in g.files:
        if key.startswith("wslice/"):
            k = key[7:]
            flat = params[k].detach().cpu().reshape(-1)
            stepv = max(1, flat.numel() // 4096)
            w0 = proc_sd[k].reshape(-1)[::stepv][:4096].double().numpy()
            d_ref = g[key].astype(np.float64) - w0
            d_got = flat[::stepv][:4096].double().numpy() - w0
            # at the fixture's lr (3e-6: the largest that keeps the random procedural weights out of dead-ReLU collapse) a backbone
            # weight moves by a few f32 ulps in three steps, so the delta is quantised: allow one ulp of the weight on top
            ulp = float(np.spacing(np.abs(g[key]).max().astype(np.float32)))
            dw[k] = (float(np.abs(d_got - d_ref).max()), float(np.abs(d_ref).max()), ulp)
    return scal, dw


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_three_steps_match_reference_fixture(golden_dir, proc_sd, prec):
    """contrast_train.py:397-399 + tool/torchutils.py:23-33 end to end, three consecutive iterations of the reference's own loop
    (oracle/make_goldens.py `multistep_golden`): pins the momentum buffer (first step buf = d, then 5e-4 * buf + d), the poly LR
    (new images, masks and lr each step), the flat-weight buffer <-> pack refresh between steps.  fp32: the 8 scalars of every
    step at 1e-4 and the weight DELTA after step 3 at 2e-3 of its maximum."""
    g = np.load(os.path.join(golden_dir, "step_S128_N3_x3.npz"))
    scal, dw = _multistep(proc_sd, g, prec)
    for s_ in range(int(g["steps"])):
        for k in SCALARS:
            ref = float(g[f"s{s_}/{k}"])
            assert abs(scal[s_][k] - ref) <= 1e-4 * max(1.0, abs(ref)), (s_, k, scal[s_][k], ref)
    bad = {k: v for k, v in dw.items() if v[0] > 2e-3 * v[1] + 1.01 * v[2]}       # (error, largest reference delta, one ulp of the weight)
    assert not bad, bad
    assert dw["fc8.weight"][1] > 100 * dw["fc8.weight"][2]                          # ... and at least the from-scratch head moves by >> 1 ulp


# bf16 (throughput) mode against the REFERENCE's fp32 fixtures.  Measured on an MI355X with scripts/measure_bf16_step.py
# (profiles/r02_bf16_deviation.json: the three one-step fixtures and the three-step fixture); every bar is 2x the measured worst case.
#   scalar: measured worst relative deviation over all fixtures and steps -> bar
BF16_SCALAR_BAR = {"loss": 8.4e-3,            # 4.2e-3
                   "loss_cls": 4.2e-3,        # 2.1e-3
                   "loss_er": 1.5e-2,         # 7.2e-3
                   "loss_ecr": 2.7e-2,        # 1.31e-2
                   "loss_nce": 2.7e-2,        # 1.31e-2 (step 1 of the three-step fixture)
                   "loss_intra_nce": 4.0e-2,  # 1.98e-2
                   "loss_cross_nce": 2.3e-2,  # 1.12e-2
                   "loss_cross_nce2": 3.0e-2}  # 1.47e-2
# weight gradients, per key group: (minimum cosine of the 4096-sample slice, maximum |norm ratio - 1|); measured worst in brackets.
# The PCM branch (f9, f8_3, f8_4) is the outlier: its gradient passes through the CAM gate of resnet38_contrast.py:41-48 (entries below the
# per-pixel maximum are zeroed — a discontinuous function of the CAM), and bf16 forward noise flips near-ties of that gate.
BF16_GRAD_BAR = {"fc8.": (0.9996, 0.002),           # [0.99981, 0.0006]
                 "fc_proj.": (0.961, 0.10),         # [0.98061, 0.0488]
                 "pcm": (0.50, 0.17),               # [0.75115, 0.0837]
                 "backbone": (0.966, 0.05)}         # [0.98334, 0.0245]


def _bf16_grad_bar(key):
    if key.startswith(("f9.", "f8_3.", "f8_4.")):
        return BF16_GRAD_BAR["pcm"]
    for k in ("fc8.", "fc_proj."):
        if key.startswith(k):
            return BF16_GRAD_BAR[k]
    return BF16_GRAD_BAR["backbone"]


@pytest.mark.parametrize("name", ["step_S160_N2", "step_S128_N3", "step_edge_S64_N3"])
def test_bf16_step_against_reference_fixture(golden_dir, proc_sd, name):
    """Throughput mode held to per-scalar RELATIVE bars against the reference's own fixtures (the round-1 test allowed 8 % of
    max(1, |ref|), under which loss_er ~ 0.007 could be wrong by 10x), plus per-key gradient cosine and norm-ratio bars."""
    from wseg_amd import synth
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
    model, opt, tr = _trainer(proc_sd, "bf16", "hip", n, seed, py_seed)
    lab = torch.from_numpy(g["label"]) if "label" in g.files else synth.synthetic_labels(n, seed)
    got = tr.step(synth.synthetic_images(n, size, seed).cuda(), lab.cuda())
    for k in SCALARS:
        ref = float(g["s/" + k])
        assert abs(float(got[k]) - ref) <= BF16_SCALAR_BAR[k] * abs(ref), (k, float(got[k]), ref)
    params = dict(model.named_parameters())
    for key in g.files:
        if not key.startswith("gslice/"):
            continue
        k = key[7:]
        flat = params[k].grad.detach().cpu().reshape(-1)
        stepv = max(1, flat.numel() // 4096)
        a, b = flat[::stepv][:4096].double().numpy(), g[key].astype(np.float64)
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
        ratio = float(params[k].grad.double().norm()) / float(g["gnorm/" + k])
        cos_min, ratio_max = _bf16_grad_bar(k)
        assert cos >= cos_min, (k, cos)
        assert abs(ratio - 1.0) <= ratio_max, (k, ratio)


def test_bf16_three_steps_against_reference_fixture(golden_dir, proc_sd):
    """Three consecutive optimizer steps in throughput mode: the same per-scalar bars at every step (the update itself — SGD on the
    f32 master weights, bf16 mirror refreshed by the same kernel — is pinned in fp32 by test_three_steps_match_reference_fixture)."""
    g = np.load(os.path.join(golden_dir, "step_S128_N3_x3.npz"))
    scal, dw = _multistep(proc_sd, g, "bf16")
    for s_ in range(int(g["steps"])):
        for k in SCALARS:
            ref = float(g[f"s{s_}/{k}"])
            assert abs(scal[s_][k] - ref) <= BF16_SCALAR_BAR[k] * abs(ref), (s_, k, scal[s_][k], ref)


def test_lookahead_prefix_dropped_when_frozen_state_changes(proc_sd):
    """The lookahead prefix (conv1a + frozen b2* with folded BNs) also depends on the frozen weights and BN buffers: a BN-buffer edit
    (as load_state_dict / a checkpoint resume makes) between `step(i, next_img1=x)` and `step(x)` must drop it (Engine.frozen_key)."""
    from wseg_amd import synth
    n, size, seed = 2, 96, 43
    imgs = [synth.synthetic_images(n, size, seed + j).cuda() for j in range(2)]
    labs = [synth.synthetic_labels(n, seed + j).cuda() for j in range(2)]
    outs = []
    for look in (False, True):
        model, opt, tr = _trainer(proc_sd, "fp32", "hip", n, seed, 4, lr=3e-6)
        model.set_dropout_masks(None)
        tr.rng_parity = False
        os.environ["WSEG_INTRA_KEY_SEED"] = "9"
        torch.manual_seed(17)
        try:
            tr.step(imgs[0], labs[0], next_img1=imgs[1] if look else None)
            with torch.no_grad():
                model.b2.bn_branch2a.running_mean.add_(0.25)                      # the frozen prefix now computes something else
            got = tr.step(imgs[1], labs[1])
            outs.append({k: float(v) for k, v in got.items()})
        finally:
            del os.environ["WSEG_INTRA_KEY_SEED"]
    for k in SCALARS:
        assert abs(outs[0][k] - outs[1][k]) <= 5e-5 * max(1.0, abs(outs[0][k])), (k, outs[0][k], outs[1][k])


@pytest.mark.parametrize("P", [4096, 1000, 300007])
def test_fused_nce_matches_the_unfused_formulation(P):
    """csrc/loss.hip nce_records + nce_fused (the product path: both views in one launch, nothing but records / dF written)
    against the unfused reference formulation nce_sims -> intra_pack -> nce_loss_grad on the same inputs: the records and dF
    bit for bit (same MFMA arithmetic), the three loss sums up to the order of the float atomics.  P = 1000: row tails;
    P = 300007: every wave of the record pass walks 4-5 tiles (its two-tiles-in-flight pipeline in steady state, odd and even trip counts)."""
    from wseg_amd import _lib as L
    dev = "cuda"
    g = torch.Generator().manual_seed(P)
    V = []
    for i in range(2):
        F = torch.randn(P, 128, generator=g)
        F[5] = 0.0                                       # a dead pixel: zero feature row (F.normalize eps path)
        V.append(dict(F=F.to(dev), p=torch.nn.functional.normalize(torch.randn(21, 128, generator=g), dim=1).to(dev),
                      y=torch.randint(0, 21, (P,), generator=g, dtype=torch.int32).to(dev),
                      w=((torch.rand(P, generator=g) < 0.4).float() * torch.rand(P, generator=g) / P).to(dev),
                      rkey=torch.rand(P, generator=g).to(dev)))
    cc, ci = 0.1 / (2 * P), 0.05
    ref_sums = torch.zeros(3, device=dev)
    for v, o in ((V[0], V[1]), (V[1], V[0])):
        v["fn"], v["nrm"] = torch.empty(P, 128, device=dev), torch.empty(P, device=dev)
        v["So"], v["St"] = torch.empty(P, 21, device=dev), torch.empty(P, 21, device=dev)
        L.nce_sims(v["F"], v["p"], o["p"], v["fn"], v["nrm"], v["So"], v["St"], P)
        v["rec_ref"] = torch.empty(3, P, device=dev)
        L.intra_pack(v["y"], v["So"], v["rkey"], v["rec_ref"], P)
        v["dF_ref"] = torch.empty(P, 128, device=dev)
        L.nce_loss_grad(v["fn"], v["nrm"], v["So"], v["St"], v["y"], o["y"], v["w"], v["p"], o["p"], v["dF_ref"], ref_sums, P, cc, ci)
    for v in V:
        v["rec"], v["dF"] = torch.full((3, P), float("nan"), device=dev), torch.full((P, 128), float("nan"), device=dev)
    L.nce_records([dict(F=v["F"], p_own=v["p"], y_own=v["y"], rkey=v["rkey"], rec=v["rec"]) for v in V], P)
    sums = torch.zeros(3, device=dev)
    L.nce_fused([dict(F=v["F"], p_own=v["p"], p_oth=o["p"], y_own=v["y"], y_oth=o["y"], w_intra=v["w"], dF=v["dF"]) for v, o in ((V[0], V[1]), (V[1], V[0]))],
                P, cc, ci, sums)
    for v in V:
        assert torch.equal(v["rec"].view(torch.int32), v["rec_ref"].view(torch.int32))
        assert torch.equal(v["dF"], v["dF_ref"])
        assert torch.isfinite(v["dF"]).all()
    assert torch.allclose(sums, ref_sums, rtol=2e-6, atol=0), (sums, ref_sums)
    assert float(ref_sums.abs().min()) > 0
    # split-bf16 form of the record pass (bf16 / bf16x3 modes): labels and keys identical, similarities to 3e-6
    rec3 = [torch.empty(3, P, device=dev) for _ in V]
    L.nce_records([dict(F=v["F"], p_own=v["p"], y_own=v["y"], rkey=v["rkey"], rec=r) for v, r in zip(V, rec3)], P, split_bf16=True)
    for v, r in zip(V, rec3):
        assert torch.equal(r[0].view(torch.int32), v["rec_ref"][0].view(torch.int32)) and torch.equal(r[2], v["rec_ref"][2])
        assert float((r[1] - v["rec_ref"][1]).abs().max()) <= 3e-6
    # the records feed the sort-based sampler through a leading dimension of 1 exactly as the [P,21] table did through 21
    if P > 8192: return                              # (the single-rank sampler sorts one view in LDS: P <= 8192)
    w21, w1 = torch.empty(P, device=dev), torch.empty(P, device=dev)
    L.intra_weights(V[0]["y"], V[0]["So"], V[0]["rkey"], None, w21, P)
    L.intra_weights(V[0]["y"], V[0]["rec"][1], V[0]["rkey"], None, w1, P, ld_s=1)
    assert torch.equal(w21, w1)


@pytest.mark.parametrize("ranks", [1, 2, 8])
def test_intra_weights_global_matches_sorted_kernel(ranks):
    """Hard-pixel sampling over the gathered batch (radix-select thresholds on the all-gathered records, one workgroup per class)
    against the single-rank sort-based kernel run on the concatenated global arrays: identical weights for every rank's slice.
    ranks = 8: the world size of BASELINE config 3 (P = 1024 per rank here: the sort-based reference kernel holds at most 8192 pixels)."""
    from wseg_amd import _lib as L
    dev = "cuda"
    P = 1536 if ranks < 8 else 1024
    g = torch.Generator().manual_seed(11 + ranks)
    PG = P * ranks
    y = torch.randint(0, 21, (PG,), generator=g, dtype=torch.int32)
    y[y == 7] = 3                                   # an absent class
    y[5] = 19; y[y == 19] = 2; y[5] = 19            # a class with a single pixel (skipped, still counted)
    S = (torch.rand(PG, 21, generator=g) * 2 - 1)
    S[10:40, :] = S[9, :]                           # tied similarities: order falls back to the pixel index
    y[10:40] = y[9]
    rk = torch.rand(PG, generator=g)
    rk[100:120] = rk[99]
    y, S, rk = y.to(dev), S.to(dev), rk.to(dev)
    w_ref = torch.empty(PG, device=dev)
    L.intra_weights(y, S, rk, None, w_ref, PG)
    rec = torch.empty(ranks, 3, P, device=dev)
    for r in range(ranks):
        sl = slice(r * P, (r + 1) * P)
        L.intra_pack(y[sl].contiguous(), S[sl].contiguous(), rk[sl].contiguous(), rec[r], P)
    for r in range(ranks):
        w = torch.empty(P, device=dev)
        L.intra_weights_global(rec, w, P, ranks, r, float(ranks), 3 * P)
        ref = w_ref[r * P:(r + 1) * P] * ranks
        assert float(ref.sum()) > 0
        assert torch.allclose(w, ref, rtol=1e-6, atol=0), float((w - ref).abs().max())


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_identical_views_property(proc_sd, prec):
    """Size-independent identity of the reference's loss: with a 128x128 input the second view is the
    first one (resize 128 -> 128 is the identity), so with the same dropout masks the two row segments of every batched
    launch carry identical data: equivariant regularisation = 0, the two ECR directions and the two cross-view NCE terms
    coincide.  Exercises the whole two-segment path (conv tiles straddling the segment boundary, prototypes, NCE)."""
    from wseg_amd import synth
    n, seed = 4, 5
    model, opt, tr = _trainer(proc_sd, prec, "hip", n, seed, 3, lr=0.0)
    m = synth.synthetic_dropout_masks(n, 77)
    model.set_dropout_masks([m, {k: v.clone() for k, v in m.items()}])
    got = tr.step(synth.synthetic_images(n, 128, seed).cuda(), synth.synthetic_labels(n, seed).cuda())
    assert float(got["loss_er"]) <= 1e-7, float(got["loss_er"])
    a, b = float(got["loss_cross_nce"]), float(got["loss_cross_nce2"])
    assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (a, b)
    for k in SCALARS:
        assert np.isfinite(float(got[k])), k


def test_full_size_step_properties(proc_sd):
    """BASELINE config 2 at its full size (B=16 x 448 x 448, bf16): properties that need no oracle — the logged scalars are
    finite and consistent (loss = cls + er + ecr + nce, nce = intra + cross + cross2), the step is repeatable (a second
    identical step with lr = 0 gives the same scalars up to the order of float atomics in the reductions), every
    trainable tensor receives a finite non-zero gradient and the frozen prefix none, and the fused SGD moves the weights."""
    from wseg_amd import synth
    n, seed = 16, 0
    model, opt, tr = _trainer(proc_sd, "bf16", "hip", n, seed, 1, lr=0.0)
    img, lab = synth.synthetic_images(n, 448, seed).cuda(), synth.synthetic_labels(n, seed).cuda()
    masks = [synth.synthetic_dropout_masks(n, 10), synth.synthetic_dropout_masks(n, 11)]
    model.set_dropout_masks([{k: v.clone() for k, v in m_.items()} for m_ in masks])
    tr.rng_parity = False
    torch.manual_seed(0)
    a = {k: float(v) for k, v in tr.step(img, lab).items()}
    for k in SCALARS:
        assert np.isfinite(a[k]), k
    assert abs(a["loss"] - (a["loss_cls"] + a["loss_er"] + a["loss_ecr"] + a["loss_nce"])) <= 1e-5 * max(1.0, abs(a["loss"]))
    assert abs(a["loss_nce"] - (a["loss_intra_nce"] + a["loss_cross_nce"] + a["loss_cross_nce2"])) <= 1e-5
    assert a["loss_er"] >= 0 and a["loss_ecr"] >= 0
    eng = model._engine
    gflat = eng.flat_g.clone()
    assert torch.isfinite(gflat).all() and float(gflat.abs().max()) > 0
    for name, (off, cnt) in eng.offsets.items():
        assert float(gflat[off:off + cnt].abs().max()) > 0, name
    assert model.conv1a.weight.grad is None and model.b2.conv_branch2a.weight.grad is None
    model.set_dropout_masks([{k: v.clone() for k, v in m_.items()} for m_ in masks])
    torch.manual_seed(0)
    b = {k: float(v) for k, v in tr.step(img, lab).items()}        # lr = 0: same weights, same inputs, same random keys
    for k in ("loss_cls", "loss_er", "loss_ecr", "loss_cross_nce", "loss_cross_nce2"):
        assert abs(a[k] - b[k]) <= 2e-6 * max(1.0, abs(a[k])), (k, a[k], b[k])
    w0 = eng.flat_w.clone()
    for g_ in opt.param_groups:
        g_["lr"] = 1e-3
    opt._PolyOptimizer__initial_lr = [1e-3 for _ in opt.param_groups]
    tr.step(img, lab)
    assert float((eng.flat_w - w0).abs().max()) > 0


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_lookahead_prefix_changes_nothing(proc_sd, prec):
    """Trainer.step(img, lab, next_img1=...) computes the frozen part of the NEXT forward pass (conv1a, b2*) on a side stream inside this step's loss
    phase.  Four steps over different batches with and without it: the 8 scalars of every step agree to the order of the float atomics in the loss sums
    and the fc8 weights after the last step to 2e-3 of their total update (the weight-gradient kernels add split-K partials with f32 atomics, so two
    runs of the SAME schedule differ by as much).  Step 3 is announced one tensor and given another (the
    announced prefix must be dropped), step 4 gets a tensor that was modified in place after it was announced."""
    from wseg_amd import synth
    n, size, seed = 2, 160, 41
    imgs = [synth.synthetic_images(n, size, seed + j).cuda() for j in range(5)]
    labs = [synth.synthetic_labels(n, seed + j).cuda() for j in range(5)]
    runs = []
    for look in (False, True):
        model, opt, tr = _trainer(proc_sd, prec, "hip", n, seed, 4, lr=3e-6)
        model.set_dropout_masks(None)
        tr.rng_parity = False
        os.environ["WSEG_INTRA_KEY_SEED"] = "9"
        torch.manual_seed(17)
        try:
            x3 = imgs[3].clone()
            plan = [(imgs[0], imgs[1]), (imgs[1], imgs[4]), (imgs[2], x3), (x3, None)]     # (step 2 -> 3: announced imgs[4], given imgs[2])
            out = []
            for j, (img, nxt) in enumerate(plan):
                if j == 3:
                    x3.mul_(0.5)                                                         # modified after it was announced
                got = tr.step(img, labs[j], next_img1=nxt if look else None)
                out.append({k: float(v) for k, v in got.items()})
        finally:
            del os.environ["WSEG_INTRA_KEY_SEED"]
        runs.append((out, model._engine.flat_w.clone()))
    (a, wa), (b, wb) = runs
    w0 = torch.cat([proc_sd[nm + ".weight"].reshape(-1) for nm in ("fc8",)]).cuda()
    o8, c8 = model._engine.offsets["fc8"]
    wa, wb = wa[o8:o8 + c8], wb[o8:o8 + c8]
    for j, (x, y) in enumerate(zip(a, b)):
        for k in SCALARS:
            if k in ("loss", "loss_nce", "loss_intra_nce", "loss_cross_nce", "loss_cross_nce2", "loss_cls", "loss_er", "loss_ecr"):
                # (step 0: float atomics in the loss sums only; later steps also carry the atomics noise of the weight gradients through the update.  bf16: that
                #  noise — 1e-7 of a weight — can flip a near-tie decision of the bf16 forward by step 3: two runs of the SAME schedule then land on one of two
                #  values 3.5e-4 apart (measured in round 3 with round 2's library too: 1.789045 / 1.789672, one run in six), so the later steps of the bf16
                #  case are held to 1e-3; the prefix being dropped or reused wrongly would move the loss by far more — it changes the whole input)
                later = 5e-5 if prec != "bf16" else 1e-3
                assert abs(x[k] - y[k]) <= (2e-6 if j == 0 else later) * max(1.0, abs(x[k])), (j, k, x[k], y[k])
    upd = float((wa - w0).abs().max())
    assert upd > 0 and float((wa - wb).abs().max()) <= 2e-3 * upd, (upd, float((wa - wb).abs().max()))


def test_packs_follow_the_weights_bf16(proc_sd):
    """After optimizer steps every derived weight buffer of the bf16 path — the cast mirror, the transposed dgrad packs (made on
    a side stream during the loss phase) and the K-concatenated packs of the two-source launches — equals what the CURRENT master
    weights give (a stale pack would train on last step's weights without any error)."""
    from wseg_amd import _lib as L, synth
    n, size, seed = 2, 96, 31
    model, opt, tr = _trainer(proc_sd, "bf16", "hip", n, seed, 3, lr=0.05)
    eng = model._engine
    for _ in range(2):
        tr.step(synth.synthetic_images(n, size, seed).cuda(), synth.synthetic_labels(n, seed).cuda())
    P = eng.ensure_packs(torch.device("cuda", torch.cuda.current_device()), L.BF16)      # (+ the deferred transposed packs)
    torch.cuda.synchronize()
    changed = 0
    for name in ("b7.conv_branch1", "b7.conv_branch2b2", "b7.conv_branch2a", "b6.conv_branch1", "b5.conv_branch2b1", "b5.conv_branch2a", "b4_2.conv_branch2a"):
        w = eng.conv_param(name).detach()                                   # logical [OC, IC, KH, KW] view of the master
        okkc = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1, w.shape[1]).to(torch.bfloat16)
        assert torch.equal(P["w"][name], okkc), name
        assert torch.equal(P["wt"][name], okkc.permute(2, 1, 0).contiguous()), name
        changed += int(not torch.equal(okkc.float(), proc_sd[name + ".weight"].permute(0, 2, 3, 1).reshape(okkc.shape).cuda().to(torch.bfloat16).float()))
    assert changed > 0                                                       # the steps did move the weights
    for blk, a, b in (("b7", "conv_branch1", "conv_branch2b2"), ("b6", "conv_branch1", "conv_branch2b2")):
        f = P["w"][blk + ".skip_fused"]
        assert torch.equal(f.reshape(f.shape[0], -1), torch.cat([P["w"][f"{blk}.{a}"].reshape(f.shape[0], -1), P["w"][f"{blk}.{b}"].reshape(f.shape[0], -1)], dim=1))
        ft = P["wt"][blk + ".skip_fused"]
        assert torch.equal(ft.reshape(ft.shape[0], -1), torch.cat([P["wt"][f"{blk}.conv_branch1"].reshape(ft.shape[0], -1),
                                                                     P["wt"][f"{blk}.conv_branch2a"].reshape(ft.shape[0], -1)], dim=1))
