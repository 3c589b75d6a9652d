#!/usr/bin/env python
"""bench.py — training images/sec of the contrast_train hot path at B=16 x 448 x 448 per GPU.

    python bench.py [--gpus N --steps K --warmup W]

N>1: one process per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: RANK / WORLD_SIZE are in the environment) or — when RANK is unset — this process starts that launcher
itself as a CHILD (before it has touched the GPU; it never re-execs) and exits with the child's return code.  A world size that
differs from --gpus is an error, so a mislaunch cannot report a one-GPU number as an N-GPU one.

One "step" = one full loop body of contrast_train.py:128-399 on a synthetic batch resident in HBM:
second view, two ResNet-38d forwards, CAM/PCM head, all SEAM + contrast losses, backward,
gradient all-reduce (N>1), PolyOptimizer step.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def conv_flops_model():
    """Algorithmic FLOPs per image of fwd+bwd (SURVEY.md §8d / BASELINE.md §2)."""
    return 2.4292e12


def cpu_baseline(sizes=(2, 16), size=448):
    """The oracle (CPU restatement, pinned to the reference) timed on this box's host cores: BASELINE.md §3 — one warm-up step,
    then one timed step (forward both views + loss + backward) at N=2 and at N=16."""
    import random
    from oracle import loss as oloss
    from oracle import net as onet
    from wseg_amd import synth

    def one(n):
        sd = synth.procedural_state_dict(0)
        for k in onet.trainable_keys(sd):
            sd[k] = sd[k].clone().requires_grad_(True)
        img = synth.synthetic_images(n, size, 0)
        lab = synth.synthetic_labels(n, 0)
        m1 = synth.synthetic_dropout_masks(n, 0)
        m2 = synth.synthetic_dropout_masks(n, 1)
        t0 = time.time()
        out = oloss.train_step(img, lab, sd, m1, m2, 0.20, random.Random(0))
        out["loss"].backward()
        return time.time() - t0

    one(1)                                                  # warm-up (thread pool, allocator, oneDNN primitive caches)
    runs = [(n, one(n)) for n in sizes]
    n, dt = runs[-1]
    return {"value": round(n / dt, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 warm-up step (N=1) then 1 timed step (fwd both views + loss + bwd) of the CPU oracle, fp32, "
                      f"{size}x{size}: " + "; ".join(f"N={n_}: {dt_:.1f} s = {n_ / dt_:.3f} img/s" for n_, dt_ in runs)
                      + f"; value = the N={n} run; nproc={os.cpu_count()}"}


def kernel_source_sha1():
    """sha1 over the conv / weight-gradient kernel sources: scripts/summarize_pmc.py stores it in the PMC traffic summary, and the bench line
    says whether the kernels of THIS build are the ones the traffic was measured on (a stale summary is then visible, not silent)."""
    import hashlib
    h = hashlib.sha1()
    for f in ("conv_igemm.hip", "conv_wgrad_kernels.h", "conv_wgrad.hip", "common.h"):
        with open(os.path.join(ROOT, "wseg_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


SIM_BYTES_RECORDS, SIM_BYTES_FUSED = 532, 1036      # algorithmic bytes per pixel (features read once; records / dF written)
HBM_PEAK_GBS = 8000.0


def similarity_section(dev):
    """The north star's second target — the pixel-to-prototype similarity (contrast_train.py:245-334) against the HBM roof — measured in
    process: `nce_records` (the similarity contraction -> hard-pixel records) and `nce_fused` (similarities + 3 InfoNCE terms + gradient),
    both views per launch, at the step's real P = 4096 pixels per view (launch-bound: SURVEY.md §8d says no bandwidth fraction is reachable there)
    and at P = 2^20 per view (the kernels' streaming rate); HIP events, SURVEY.md §8d's algorithmic bytes / time."""
    from wseg_amd import _lib as L

    def run(P, iters):
        g = torch.Generator(device=dev).manual_seed(P)
        V = []
        for _ in range(2):
            F = torch.randn(P, 128, device=dev, generator=g)
            V.append(dict(F=F, p=torch.nn.functional.normalize(torch.randn(21, 128, device=dev, generator=g), dim=1),
                          y=torch.randint(0, 21, (P,), device=dev, dtype=torch.int32, generator=g), w=torch.rand(P, device=dev, generator=g) / P,
                          dF=torch.empty_like(F), rkey=torch.rand(P, device=dev, generator=g), rec=torch.empty(3, P, device=dev)))
        sums = torch.zeros(3, device=dev)
        rv = [dict(F=v["F"], p_own=v["p"], y_own=v["y"], rkey=v["rkey"], rec=v["rec"]) for v in V]
        fv = [dict(F=v["F"], p_own=v["p"], p_oth=o["p"], y_own=v["y"], y_oth=o["y"], w_intra=v["w"], dF=v["dF"]) for v, o in ((V[0], V[1]), (V[1], V[0]))]

        def t(fn):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3       # us
        out = {}
        for name, fn, bpp, protos in (("records", lambda: L.nce_records(rv, P, split_bf16=True), SIM_BYTES_RECORDS, 1),
                                      ("records_f32", lambda: L.nce_records(rv, P), SIM_BYTES_RECORDS, 1),
                                      ("fused", lambda: L.nce_fused(fv, P, 0.1 / (2 * P), 0.05, sums), SIM_BYTES_FUSED, 2)):
            us = t(fn)
            nbytes = 2 * (P * bpp + protos * 21 * 128 * 4)
            out[name] = {"us": round(us, 2), "GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}
        return out
    return {"bound": "hbm", "peak_GBps": HBM_PEAK_GBS, "bytes_per_pixel": {"records": SIM_BYTES_RECORDS, "fused": SIM_BYTES_FUSED},
            "note": "both views per launch; `records` = the similarity contraction of the bf16 / bf16x3 modes (split-bf16 products), `records_f32` = exact-f32 MFMA (fp32 mode), "
                    "`fused` = similarities + InfoNCE + gradient (exact f32 in every mode)",
            "P4096_real_shape": run(4096, 50), "P1048576_sweep": run(1 << 20, 3)}


def infer_section(dev, n_img=8):
    """BASELINE config 5's geometry (contrast_infer.py:49-99: one 375 x 500 image, scales 0.5 / 1 / 1.5 / 2 x flip = 8 forwards, post-process) on
    inputs resident in HBM, bf16 mode: images/s, ms/image and the fraction of the dense bf16 MFMA peak over the algorithmic FLOPs of the 8
    forwards (wseg_amd.arch.forward_macs: SURVEY.md §8d's counting)."""
    import torch.nn.functional as F
    from wseg_amd import arch, synth
    from wseg_amd.infer import infer_image
    from wseg_amd.resnet38_contrast import Net
    H, W = 375, 500
    model = Net(precision="bf16")
    model.load_state_dict(synth.procedural_state_dict(0, device=dev))
    model.eval(); model.cuda(dev)
    g = torch.Generator().manual_seed(0)
    base = torch.randn(1, 3, H, W, generator=g).to(dev)
    label = torch.zeros(20); label[[3, 11]] = 1
    lst, flops = [], 0.0
    for s_ in (0.5, 1.0, 1.5, 2.0):
        hs, ws = int(round(H * s_)), int(round(W * s_))
        im = F.interpolate(base, size=(hs, ws), mode="bicubic", align_corners=False)
        lst += [im, im.flip(-1)]
        flops += 2 * 2.0 * arch.forward_macs(hs, ws)
    for _ in range(2):
        infer_image(model, lst, label, (H, W))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_img):
        infer_image(model, lst, label, (H, W))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_img
    return {"workload": "contrast_infer multi-scale CAM, one synthetic 375x500 image = 8 forwards (scales 0.5/1/1.5/2 x flip) + post-process, inputs resident, bf16",
            "images_per_sec": round(1.0 / dt, 2), "ms_per_image": round(dt * 1e3, 2), "forwards_per_image": 8, "images_timed": n_img,
            "algorithmic_tflop_per_image": round(flops / 1e12, 3), "achieved_tflops": round(flops / dt / 1e12, 1),
            "frac_of_bf16_peak": round(flops / dt / 1e12 / PEAK_BF16_TFLOPS, 4), "val_set_1449_images_sec": round(1449 * dt, 1)}


def _spawn_ranks(a):
    """--gpus N without a launcher: start `torch.distributed.run` as a child process (this process has made no GPU call)."""
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode          # rank 0's JSON line goes to the inherited stdout


def timed_steps(trainer, batches, steps, warmup, barrier, lookahead=True):
    """batches: two (images, labels) pairs used alternately; step i is told the images of step i + 1 (as a prefetching loader would),
    so the frozen prefix of the next forward pass runs inside the current step's loss phase."""
    def one(i):
        img, lab = batches[i % 2]
        return trainer.step(img, lab, next_img1=batches[(i + 1) % 2][0] if lookahead else None)
    for i in range(warmup):
        one(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        losses = one(i)
    barrier()
    return time.perf_counter() - t0, losses


def build_trainer(precision, lr, dev, rank):
    import random
    from wseg_amd import synth
    from wseg_amd.optim import PolyOptimizer
    from wseg_amd.resnet38_contrast import Net
    from wseg_amd.train import Trainer
    model = Net(precision=precision)
    groups = _quiet(model.get_parameter_groups)
    opt = PolyOptimizer([
        {'params': groups[0], 'lr': lr, 'weight_decay': 5e-4},
        {'params': groups[1], 'lr': 2 * lr, 'weight_decay': 0},
        {'params': groups[2], 'lr': 10 * lr, 'weight_decay': 5e-4},
        {'params': groups[3], 'lr': 20 * lr, 'weight_decay': 0}], lr=lr, weight_decay=5e-4, max_step=10582 // 16 * 8)
    model.load_state_dict(synth.procedural_state_dict(0, device=dev))
    model.cuda(dev)
    model.train()
    return model, Trainer(model, opt, 0.20, random.Random(1000 + rank), False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the `similarity` (HBM roof of the pixel-to-prototype kernels) and `infer_mode` (config 5 geometry) sections")
    ap.add_argument("--cpu-baseline-n", default="2,16", help="batch sizes of the CPU-oracle leg (BASELINE.md §3: N=2 and N=16)")
    ap.add_argument("--parity-steps", type=int, default=2,
                    help="timed steps of the parity (f32-exact) mode reported as `parity_mode` beside the bf16 line (0: skip)")
    ap.add_argument("--parity-precision", default=os.environ.get("WSEG_PARITY_PRECISION", "fp32,bf16x3"),
                    help="comma list of the parity-grade modes to time: fp32 (exact-f32 MFMA), bf16x3 (f32 storage, split-bf16 products)")
    ap.add_argument("--event-stride", type=int, default=5, help="bracket every k-th conv launch with HIP events (rotating)")
    ap.add_argument("--no-lookahead", action="store_true", help="do not tell step i the images of step i + 1 (A/B switch: the frozen prefix of the "
                                                                "next forward pass then runs in front of it instead of inside the loss phase)")
    ap.add_argument("--seed", type=int, default=0, help="base seed: rank r draws its images / labels / dropout masks / keys from seed + r")
    ap.add_argument("--lr", type=float, default=1e-5,
                    help="base lr; the reference's 0.01 makes the RANDOM procedural weights diverge within 2 steps "
                         "(measured, scripts/debug_step.py), so the bench steps with a small lr — same kernels, same work")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:              # no launcher above us: be the launcher (child process, no exec)
        sys.exit(_spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch `python bench.py --gpus N` (it starts the ranks) "
                         f"or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # WSEG_DIST_BACKEND=gloo: protocol rehearsal with several ranks on ONE card (RCCL refuses two ranks per device)
    backend = os.environ.get("WSEG_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.manual_seed(a.seed + rank)                          # Dropout2d masks and hard-pixel keys differ per rank (they see different images)
    import torch.distributed as dist
    force_dist = os.environ.get("WSEG_FORCE_DIST", "0") == "1"     # one-rank RCCL group: exercises the exchange code on one GPU
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when the communicator is created: this process's stdout carries ONE JSON line, so file
        # descriptor 1 points at stderr while the group is set up (and the C stdio buffer is flushed before it is restored)
        import ctypes
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
            ctypes.CDLL(None).fflush(None)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    from wseg_amd import _lib as L
    from wseg_amd import synth

    model, trainer = build_trainer(a.precision, a.lr, dev, a.seed + rank)
    # two synthetic batches used alternately: step i is given the images of step i + 1, as a prefetching data loader would
    batches = [(synth.synthetic_images(a.batch, a.size, seed=a.seed + rank + 7919 * j, device=dev),
                synth.synthetic_labels(a.batch, seed=a.seed + rank + 7919 * j, device=dev)) for j in range(2)]
    look = not a.no_lookahead

    def one_step(i):
        img, lab = batches[i % 2]
        return trainer.step(img, lab, next_img1=batches[(i + 1) % 2][0] if look else None)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_step(i)
    # HIP events around the conv_igemm launches, live in the timed region, on the launch stream.  An event pair costs
    # ~10 us of stream time, so launch i of step s is bracketed only when (i + s) % stride == 0: every launch index
    # is sampled steps/stride times while the timed region slows by ~0.2 ms/step instead of ~1 ms.
    L.PROFILE = []
    L.PROFILE_STRIDE = max(1, min(a.event_stride, a.steps))
    barrier()
    t0 = time.perf_counter()
    for it in range(a.steps):
        L.profile_begin_step(it)
        losses = one_step(a.warmup + it)
    barrier()
    dt = time.perf_counter() - t0
    prof, L.PROFILE = L.PROFILE, None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt
    PEAKS = {"bf16": PEAK_BF16_TFLOPS, "fp32": PEAK_F32_TFLOPS, "bf16x3": PEAK_BF16_TFLOPS / 3}

    if rank == 0:
        # dominant kernel: the implicit-GEMM conv (fwd + dgrad launches), timed with HIP events on the
        # launch stream during the timed region
        per_idx = {}                                           # launch index within a step -> [sum ms, samples, flops]
        for p_ in prof:
            e = per_idx.setdefault(p_[4], [0.0, 0, p_[2]])
            e[0] += p_[0].elapsed_time(p_[1]); e[1] += 1
        tot_ms = sum(e[0] / e[1] for e in per_idx.values()) * a.steps     # = one step's launches, averaged over their samples
        tot_fl = sum(e[2] for e in per_idx.values()) * a.steps
        n_launch = max(1, len(per_idx)) * a.steps
        peak = PEAKS[a.precision]
        achieved = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        traffic, traffic_src, traffic_fresh = None, None, None
        try:                                                   # HBM bytes per launch from the committed PMC passes
            import glob                                        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 on gfx950)
            f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
            pj = json.load(open(f))
            traffic = round(pj["conv_igemm"]["hbm_bytes_per_launch"])
            # the summary names the kernel sources it was measured on (scripts/summarize_pmc.py): a summary older than the kernels says so here
            traffic_src = {"file": os.path.relpath(f, ROOT), "measured_on_kernel_src_sha1": pj.get("_kernel_src_sha1"), "measured_at": pj.get("_measured_at")}
            traffic_fresh = pj.get("_kernel_src_sha1") == kernel_source_sha1()
        except Exception:
            pass
        roof = {"bound": "mfma", "kernel": "conv_igemm256_kernel + conv_bwd_pair_kernel + conv_igemm512x128_kernel + conv_igemm_kernel (every fwd / dgrad conv launch; a dgrad launch that carries its layer's weight-gradient tiles counts their flops too)", "achieved": round(achieved, 1), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src, "traffic_matches_this_build": traffic_fresh,
                "launches_per_step": n_launch // a.steps, "event_samples": len(prof),
                "avg_launch_ms": round(tot_ms / n_launch, 4), "avg_launch_gflop": round(tot_fl / n_launch / 1e9, 2),
                "whole_step_frac": round(conv_flops_model() * a.batch / (ms * 1e-3) / 1e12 / peak, 4)}
        line = {"metric": "training images/sec at B=16x448x448 (contrast_train step)", "value": round(value, 2),
                "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": {"bf16": "bf16", "fp32": "f32", "bf16x3": "bf16x3 (hi/lo split, f32 accumulate)"}[a.precision], "data": "synthetic",
                "config": {"workload": f"ResNet-38 contrast, synthetic VOC {a.size}x{a.size}, B={a.batch}/GPU, "
                                       f"procedural weights, dropout on, fused HIP loss",
                           "global_batch": a.batch * world, "parallelism": f"dp{world}", "seed": a.seed,
                           "input": "two synthetic batches alternate" + ("; step i is given the images of step i+1 (prefetching-loader lookahead)" if look else "")},
                "loss": float(losses["loss"]), "roofline": roof}
    # ---- the parity mode's throughput (the mode the 1e-4 / argmax-exact evidence is for), same workload, rank 0 at N=1 only
    if rank == 0 and world == 1 and a.parity_steps > 0 and a.precision == "bf16":
        del trainer, model
        torch.cuda.empty_cache()
        line["parity_mode"] = []
        for pprec in a.parity_precision.split(","):
            pmodel, ptrainer = build_trainer(pprec, a.lr, dev, a.seed + rank)
            pdt, plosses = timed_steps(ptrainer, batches, a.parity_steps, 1, barrier, look)
            pms = pdt / a.parity_steps * 1e3
            ppeak = PEAKS[pprec]
            line["parity_mode"].append({"precision": pprec, "ms_per_step": round(pms, 2), "value": round(a.batch * a.parity_steps / pdt, 2),
                                        "unit": "images/sec", "steps": a.parity_steps, "warmup": 1, "peak_tflops": round(ppeak, 1),
                                        "whole_step_frac": round(conv_flops_model() * a.batch / (pms * 1e-3) / 1e12 / ppeak, 4),
                                        "loss": float(plosses["loss"])})
            del ptrainer, pmodel
            torch.cuda.empty_cache()
    # ---- the rest of the north star's evidence, rank 0 at N=1 only, after the training measurement: the similarity kernels against the HBM
    #      roof and BASELINE config 5's inference geometry
    if rank == 0 and world == 1 and not a.no_extras:
        line["similarity"] = similarity_section(dev)
        line["infer_mode"] = infer_section(dev)
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(tuple(int(x) for x in a.cpu_baseline_n.split(",")))
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def _quiet(fn):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn()


if __name__ == "__main__":
    main()
