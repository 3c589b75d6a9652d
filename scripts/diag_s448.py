"""Which discrete decision differs between the HIP fp32 step and the reference on the S448 fixture (loss_nce off by 1.1e-4)?"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_loss import _trainer, SCALARS
from wseg_amd import synth
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "step_S448_N2.npz"))
n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
sd = synth.procedural_state_dict(0)
model, opt, tr = _trainer(sd, prec, "hip", n, seed, py_seed)
model._engine.capture_ctx = True
got = tr.step(synth.synthetic_images(n, size, seed).cuda(), synth.synthetic_labels(n, seed).cuda())
for k in SCALARS:
    print(f"{k:18s} {float(got[k]):.7f} ref {float(g['s/' + k]):.7f} diff {float(got[k]) - float(g['s/' + k]):+.2e}")
v1, v2 = model._engine.last_loss_views
for nm, v, py, pp in (("view1", v1, g["pseudo1"], g["protos1"]), ("view2", v2, g["pseudo2"], g["protos2"])):
    y = v.y.cpu().numpy()
    bad = np.nonzero(y != py.astype(np.int32))[0]
    print(nm, "pseudo-label mismatches:", len(bad), bad[:10], "hip", y[bad[:10]], "ref", py[bad[:10]])
    d = np.abs(v.protos.cpu().numpy() - pp)
    print(nm, "prototype max abs diff per class:", np.round(d.max(axis=1), 6))
