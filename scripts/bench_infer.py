"""Throughput of the multi-scale CAM inference (BASELINE config 5: contrast_infer.py over VOC-sized images, scales
0.5/1/1.5/2 + flips = 8 forwards per image) on synthetic 500x375 inputs, resident on the device."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from wseg_amd import synth
from wseg_amd.infer import infer_image
from wseg_amd.resnet38_contrast import Net

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = "cuda"
model = Net(precision=prec)
model.load_state_dict(synth.procedural_state_dict(0))
model.eval(); model.cuda()
H, W = 375, 500
g = torch.Generator().manual_seed(0)
base = torch.randn(1, 3, H, W, generator=g).to(dev)
label = torch.zeros(20); label[[3, 11]] = 1
lst = []
for s in (0.5, 1.0, 1.5, 2.0):
    im = F.interpolate(base, size=(int(round(H * s)), int(round(W * s))), mode="bicubic", align_corners=False)
    lst += [im, im.flip(-1)]
for _ in range(2):
    infer_image(model, lst, label, (H, W))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_img):
    norm_cam, pred, cam_dict = infer_image(model, lst, label, (H, W))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"inference {prec}: {n_img / dt:.2f} images/s ({dt / n_img * 1e3:.1f} ms per image, 8 inputs = 2 two-segment launch sequences of batch-of-two, {H}x{W}); "
      f"1449 val images would take {1449 * dt / n_img:.0f} s")
