"""Stem (conv1a 3->64 + BN-ReLU, NCHW f32 -> NHWC bf16) at the 448x448 view: scalar-FMA kernel vs packed-FMA kernel."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wseg_amd import _lib as L
x = torch.randn(16, 3, 448, 448, device="cuda"); w = torch.randn(64, 27, device="cuda") * 0.1
sc = torch.rand(64, device="cuda") + 0.5; sh = torch.randn(64, device="cuda")
outs = []
for name, fn, wt in (("scalar", L.stem_conv, w), ("packed", L.stem_conv_kc, w.t().contiguous())):
    act = torch.empty(16 * 448 * 448, 64, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): fn(x, wt, sc, sh, None, act, 16, 448, 448, L.BF16)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn(x, wt, sc, sh, None, act, 16, 448, 448, L.BF16)
    e.record(); torch.cuda.synchronize()
    outs.append(act)
    print(f"stem {name}: {s.elapsed_time(e)/20*1e3:.1f} us per launch ({act.numel()*2/(s.elapsed_time(e)/20*1e-3)/1e12:.2f} TB/s of output)")
print("bit-identical outputs:", bool((outs[0] == outs[1]).all()))
