"""BASELINE config 5 end to end: `python -m wseg_amd.contrast_infer` (the counterpart of contrast_infer.py) over a synthetic val set of VOC
geometry — N JPEG files of 500 x 375 / 375 x 500 (as VOC's landscape / portrait mix), 1-3 labels per image — from JPEG files to the arg-max pngs
(--out_cam_pred) and, for a subset, the `<name>.npy` CAM dictionaries.  Wall time includes decoding, the eight host-side resizes per image in the
loader workers, the device work and the png writes.
  python scripts/bench_infer_cli.py [N=1449] [precision=bf16] [workers=16]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, PIL.Image, torch
from wseg_amd import contrast_infer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1449
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 16
tmp = tempfile.mkdtemp(prefix="wseg_val_")
try:
    root = os.path.join(tmp, "VOC2012"); os.makedirs(os.path.join(root, "JPEGImages"))
    rng = np.random.default_rng(5)
    names, labels = [], {}
    t0 = time.perf_counter()
    for i in range(N):
        n = f"2008_{i:06d}"
        hw = (375, 500) if i % 4 else (500, 375)
        low = rng.integers(0, 256, (hw[0] // 25, hw[1] // 25, 3), dtype=np.uint8)         # smooth content: a low-resolution random image, upsampled
        PIL.Image.fromarray(low).resize((hw[1], hw[0]), PIL.Image.BICUBIC).save(os.path.join(root, "JPEGImages", n + ".jpg"), quality=90)
        lab = np.zeros(20, np.float32); lab[rng.choice(20, size=int(rng.integers(1, 4)), replace=False)] = 1
        names.append(n); labels[n] = lab
    lst = os.path.join(tmp, "val.txt")
    open(lst, "w").write("\n".join(f"/JPEGImages/{n}.jpg /SegmentationClassAug/{n}.png" for n in names) + "\n")
    np.save(os.path.join(tmp, "cls_labels.npy"), labels, allow_pickle=True)
    print(f"{N} synthetic val JPEGs written in {time.perf_counter() - t0:.1f} s", flush=True)
    warm = os.path.join(tmp, "warm.txt")
    open(warm, "w").write("\n".join(f"/JPEGImages/{n}.jpg /SegmentationClassAug/{n}.png" for n in names[:8]) + "\n")
    common = ["--weights", "procedural", "--voc12_root", root, "--labels", os.path.join(tmp, "cls_labels.npy"), "--precision", prec]
    contrast_infer.main(common + ["--infer_list", warm, "--out_cam_pred", os.path.join(tmp, "warm_pred"), "--num_workers", "2"])   # (first import / packs / allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    contrast_infer.main(common + ["--infer_list", lst, "--out_cam_pred", os.path.join(tmp, "pred"), "--num_workers", str(workers)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    npng = len(os.listdir(os.path.join(tmp, "pred")))
    print(f"contrast_infer [{prec}], {N} val images (500x375 / 375x500), 4 scales x flip, {workers} loader workers: {dt:.1f} s wall = {N / dt:.1f} images/s "
          f"(model load + packs included; {npng} pngs written)", flush=True)
    def timed(names_, tag, extra):
        f = os.path.join(tmp, tag + ".txt")
        open(f, "w").write("\n".join(f"/JPEGImages/{n}.jpg /SegmentationClassAug/{n}.png" for n in names_) + "\n")
        t0 = time.perf_counter()
        contrast_infer.main(common + ["--infer_list", f, "--num_workers", str(workers)] + extra)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    # start-up (procedural weights, packs, forking the loader workers) is the same for any list length: the slope between two lengths is the steady state
    t208 = timed(names[:208], "s208", ["--out_cam_pred", os.path.join(tmp, "p208")])
    print(f"  208 images: {t208:.1f} s -> steady state, pngs only: {(N - 208) / (dt - t208):.1f} images/s (start-up ~{t208 - 208 * (dt - t208) / (N - 208):.1f} s)", flush=True)
    c208 = timed(names[:208], "c208", ["--out_cam", os.path.join(tmp, "c208"), "--out_cam_pred", os.path.join(tmp, "q208")])
    c608 = timed(names[:608], "c608", ["--out_cam", os.path.join(tmp, "c608"), "--out_cam_pred", os.path.join(tmp, "q608")])
    size = sum(os.path.getsize(os.path.join(tmp, "c208", f)) for f in os.listdir(os.path.join(tmp, "c208")))
    print(f"  steady state with --out_cam as well (the <name>.npy CAM dictionaries, {size / 208e6:.2f} MB per image): {400 / (c608 - c208):.1f} images/s", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
