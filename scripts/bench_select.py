"""Times wseg_select_kth (radix select of the k-th order statistic per row) at the two shapes of the training step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import _lib as L
dev = "cuda"
def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for name, rows, n, k, largest, use_abs in (("ecr", 32, 21 * 128 * 128, int(21 * 128 * 128 * 0.2), True, True), ("rvmin", 16, 448 * 448, 448 * 448 // 4, False, False)):
    for dist in ("normal*0.05", "uniform"):
        v = (torch.randn(rows, n, device=dev) * 0.05) if dist.startswith("normal") else torch.rand(rows, n, device=dev)
        if name == "rvmin": v = v.abs()
        ws = torch.empty(L.select_workspace_bytes(rows), device=dev, dtype=torch.uint8)
        res = torch.empty(rows, 4, device=dev)
        t = timeit(lambda: L.select_kth(v, rows, n, k, largest, use_abs, False, res, ws))
        ref = (v.abs() if use_abs else v).topk(k, dim=1, largest=largest)[0][:, -1]
        ok = torch.equal(res[:, 0], ref)
        print(f"{name:6s} rows {rows} n {n} {dist:12s} gx={os.environ.get('WSEG_SELECT_GX','auto')}: {t:7.1f} us per select_kth (4 hist + 4 scan + sum), threshold exact: {ok}", flush=True)
