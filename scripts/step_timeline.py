"""Timeline of the non-conv part of ONE steady-state training step from a rocprofv3 --kernel-trace csv: every kernel that is not a conv / wgrad launch,
with start / end relative to the step's first kernel, its queue, and the gaps; conv launches are collapsed into runs.
  python scripts/step_timeline.py <run_kernel_trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
sg = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
a, b = sg[-2], sg[-1]
seg = rows[a + 1:b + 1]
t0 = seg[0]["s"]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n) if not n.startswith("at::") else re.sub(r"<.*$", "", n)
    return n[:60]
run = None
prev_end = t0
for r in seg:
    k = short(r["Kernel_Name"])
    heavy = k.startswith("conv_igemm") or k.startswith("conv_wgrad") or k.startswith("conv_bwd_pair") or k.startswith("wseg_wg::conv_wgrad")
    if heavy:
        if run is None: run = [r["s"], r["e"], 1]
        else: run[1] = max(run[1], r["e"]); run[2] += 1
        continue
    if run is not None:
        print(f"{(run[0]-t0)/1e3:10.1f} {(run[1]-t0)/1e3:10.1f}  ---- {run[2]} conv/wgrad launches ({(run[1]-run[0])/1e3:.1f} us)")
        run = None
    print(f"{(r['s']-t0)/1e3:10.1f} {(r['e']-t0)/1e3:10.1f}  q{r.get('Queue_Id','?'):>2s} {(r['e']-r['s'])/1e3:8.1f} us  {k}")
if run is not None: print(f"{(run[0]-t0)/1e3:10.1f} {(run[1]-t0)/1e3:10.1f}  ---- {run[2]} conv/wgrad launches ({(run[1]-run[0])/1e3:.1f} us)")
