"""bf16 (throughput) mode against the REFERENCE's fp32 fixtures: per-scalar relative deviation, and per gradient key the cosine
of the 4096-sample slice and the norm ratio — the measurements the per-scalar bars of tests/test_gpu_loss.py are set from
(bars = 2x these).  Also the 3-step fixture.  Prints one JSON object.

    python scripts/measure_bf16_step.py [fp32|bf16|bf16x3]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_gpu_loss import SCALARS, _trainer, _multistep
from wseg_amd import synth

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sd = synth.procedural_state_dict(0)
out = {}
for name in ("step_S160_N2", "step_S128_N3", "step_edge_S64_N3", "step_S448_N2", "step_S448_N2_b"):
    g = np.load(os.path.join(G, name + ".npz"))
    n, size, seed, py_seed = int(g["n"]), int(g["size"]), int(g["seed"]), int(g["py_seed"])
    model, opt, tr = _trainer(sd, prec, "hip", n, seed, py_seed)
    lab = torch.from_numpy(g["label"]) if "label" in g.files else synth.synthetic_labels(n, seed)
    got = tr.step(synth.synthetic_images(n, size, seed).cuda(), lab.cuda())
    r = {"scalars": {k: (float(got[k]), float(g["s/" + k]), abs(float(got[k]) - float(g["s/" + k])) / max(1e-12, abs(float(g["s/" + k])))) for k in SCALARS}}
    params = dict(model.named_parameters())
    gr = {}
    for key in g.files:
        if key.startswith("gslice/"):
            k = key[7:]
            flat = params[k].grad.detach().cpu().reshape(-1)
            stepv = max(1, flat.numel() // 4096)
            a, b = flat[::stepv][:4096].double().numpy(), g[key].astype(np.float64)
            gr[k] = {"cos": float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300)),
                     "norm_ratio": float(params[k].grad.double().norm()) / float(g["gnorm/" + k]),
                     "max_rel": float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))}
    r["grads"] = gr
    out[name] = r
g = np.load(os.path.join(G, "step_S128_N3_x3.npz"))
scal, dw = _multistep(sd, g, prec)
out["step_S128_N3_x3"] = {"scalars": {f"s{s}/{k}": (scal[s][k], float(g[f"s{s}/{k}"]), abs(scal[s][k] - float(g[f"s{s}/{k}"])) / max(1e-12, abs(float(g[f"s{s}/{k}"]))))
                                      for s in range(int(g["steps"])) for k in SCALARS}, "delta_w (max error, max reference delta, one ulp)": dw}
print(json.dumps({"precision": prec, "results": out}))
