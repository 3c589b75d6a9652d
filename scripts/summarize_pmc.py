"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into HBM bytes per launch per kernel
family — the `traffic` figure of bench.py's roofline object.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <out>/pmc_fetch -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <out>/pmc_write -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python scripts/summarize_pmc.py <out>/pmc_fetch/run_counter_collection.csv <out>/pmc_write/run_counter_collection.csv profiles/rNN_pmc_traffic.json

Units / corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of wide (16 B / lane) coalesced reads — doubled here; WRITE_SIZE is exact
for 16-B-per-lane stores and float atomics.  Calibration: `sgd` must come out at ~5 x 420 MB per launch.
"""
import collections
import csv
import datetime
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FAMILIES = [("conv_bwd_pair", "conv_bwd_pair"), ("conv_igemm256", "conv_igemm256"), ("conv_igemm512", "conv_igemm512"), ("conv_igemm_kernel", "conv_igemm128"), ("conv_wgrad", "conv_wgrad"),
            ("sgd_kernel", "sgd"), ("stem_kernel", "stem"), ("pcm_", "pcm"), ("nce_", "nce"), ("up_", "maps"),
            ("pack_tr", "pack"), ("to_bf16", "pack")]


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return "other"


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = agg[family(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"]) * 1024.0
    return agg


def main():
    fetch, write, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fa, wa = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 3 --warmup 1`; bytes = "
                    "counter*1024; FETCH doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); "
                    "calibration: sgd = 3 reads + 2(+1 bf16 mirror) writes of the 420 MB flat buffers"}
    from bench import kernel_source_sha1                    # the kernels this traffic belongs to: bench.py flags a summary older than its kernels
    res["_kernel_src_sha1"] = kernel_source_sha1()
    res["_measured_at"] = datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ")
    conv = [0, 0.0, 0.0]
    for fam in sorted(set(fa) | set(wa)):
        n = max(fa[fam][0], wa[fam][0])
        if n == 0:
            continue
        fb, wb = 2.0 * fa[fam][1] / max(1, fa[fam][0]), wa[fam][1] / max(1, wa[fam][0])
        res[fam] = {"launches": n, "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
        if fam in ("conv_igemm256", "conv_igemm512", "conv_igemm128", "conv_bwd_pair"):       # (bench.py's roofline launches: fwd, dgrad, dgrad+wgrad grids)
            conv[0] += n; conv[1] += fb * n; conv[2] += wb * n
    if conv[0]:
        res["conv_igemm"] = {"launches": conv[0], "fetch_bytes_per_launch_corrected": conv[1] / conv[0],
                             "write_bytes_per_launch": conv[2] / conv[0], "hbm_bytes_per_launch": (conv[1] + conv[2]) / conv[0]}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:16s} launches {v['launches']:5d}  fetch {v['fetch_bytes_per_launch_corrected']/1e6:9.1f} MB  "
                  f"write {v['write_bytes_per_launch']/1e6:9.1f} MB  per launch")


if __name__ == "__main__":
    main()
