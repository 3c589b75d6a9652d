"""Steady-state breakdown of one training step from a rocprofv3 --kernel-trace csv: the kernels between two consecutive sgd_kernel launches
(the last full steps of the run), grouped by name, with the wall time of the step, the busy time per stream (queue) and the idle gaps.
  python scripts/step_breakdown.py <run_kernel_trace.csv> [n_last_steps]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
sg = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
assert len(sg) > nlast, len(sg)
a, b = sg[-nlast - 1], sg[-1]
seg = rows[a + 1:b + 1]
wall = (rows[b]["e"] - rows[a]["e"]) / nlast
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n) if not n.startswith("at::") else re.sub(r"<.*$", "", n)
    return n[:70]
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = short(r["Kernel_Name"]); agg[k][0] += 1; agg[k][1] += r["e"] - r["s"]
# union of busy intervals
iv = sorted((r["s"], r["e"]) for r in seg); busy = 0; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(v[1] for v in agg.values())
print(f"steps analysed: {nlast}; wall {wall/1e6:.3f} ms/step; sum of kernel durations {tot/nlast/1e6:.3f} ms/step; union busy {busy/nlast/1e6:.3f} ms/step; idle {(wall*nlast-busy)/nlast/1e6:.3f} ms/step")
q = collections.Counter()
for r in seg: q[r.get("Queue_Id", "?")] += r["e"] - r["s"]
print("per queue busy ms/step:", {k: round(v / nlast / 1e6, 3) for k, v in q.items()})
groups = collections.OrderedDict([("conv fwd / dgrad (+ paired wgrad)", ("conv_igemm", "conv_bwd_pair")), ("wgrad alone", ("conv_wgrad", "wseg_wg::conv_wgrad")), ("copies/fills (rocclr)", ("__amd_rocclr",)), ("torch elementwise etc.", ("at::",))])
gs = collections.Counter()
for k, (c, t) in agg.items():
    g = next((g for g, pats in groups.items() if any(k.startswith(p) for p in pats)), "other hand-written")
    gs[g] += t
print("groups ms/step:", {k: round(v / nlast / 1e6, 3) for k, v in gs.items()})
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{k:72s} {c/nlast:7.1f} calls {t/nlast/1e3:9.1f} us/step")
