"""MFMA pipe utilisation and effective clock per kernel family from a counters-only rocprofv3 pass:
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -o run -- python bench.py --steps 3 --warmup 1 ...
  python scripts/summarize_mfma.py DIR/.../run_counter_collection.csv DIR/.../run_kernel_trace.csv out.json
SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so chip cycles = GUI / 8, effective clock =
GUI / 8 / kernel time and mfma_pipe_busy = MFMA_BUSY / (1024 * GUI / 8)  (MI355X_MICROARCH.md)."""
import csv, json, sys, collections
FAMILIES = [("conv_bwd_pair", "conv_bwd_pair (dgrad + wgrad tiles in one grid)"), ("conv_igemm256", "conv_igemm256"), ("conv_igemm512", "conv_igemm512"),
            ("conv_igemm_kernel", "conv_igemm128"), ("conv_wgrad_pipe", "conv_wgrad_pipe"), ("conv_wgrad_kernel", "conv_wgrad128"), ("pcm_", "pcm"), ("nce_", "nce")]
def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return None
cc, kt, out = sys.argv[1:4]
dur = {}
for r in csv.DictReader(open(kt, newline="")):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: {"launches": set(), "mfma": 0.0, "gui": 0.0})
for r in csv.DictReader(open(cc, newline="")):
    fam = family(r["Kernel_Name"])
    if fam is None:
        continue
    a = agg[fam]
    a["launches"].add(r["Dispatch_Id"])
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": a["mfma"] += float(r["Counter_Value"])
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE": a["gui"] += float(r["Counter_Value"])
res = {"_note": __doc__.split("\n")[0] + " — counters only (no other trace domain); profiled passes run slower than the un-profiled bench."}
for fam, a in agg.items():
    t_ns = sum(dur.get(d, 0) for d in a["launches"])
    chip = a["gui"] / 8.0
    if chip <= 0 or t_ns <= 0:
        continue
    res[fam] = {"launches": len(a["launches"]), "mfma_pipe_busy": round(a["mfma"] / (1024.0 * chip), 4),
                "effective_clock_ghz": round(chip / t_ns, 3), "kernel_ms_total": round(t_ns / 1e6, 2)}
json.dump(res, open(out, "w"), indent=1)
for k, v in res.items():
    if k != "_note": print(k, v)
