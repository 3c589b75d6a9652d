#!/bin/bash
# Runs ON the GPU box (through gpurun): PMC traffic passes, rocprofv3 kernel stats, per-layer profile and the bench line of the
# current build; everything lands under gpurun_out/refresh/ (copy what should be judged into profiles/ afterwards).
#   scripts/gpu.sh 1100 'bash scripts/refresh_profiles.sh'
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/refresh
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o run -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o run -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
python "$ROOT/scripts/summarize_pmc.py" $(find "$OUT/pmc_fetch" -name run_counter_collection.csv) $(find "$OUT/pmc_write" -name run_counter_collection.csv) "$OUT/pmc_traffic.json" > "$OUT/pmc_summary.txt"
cp "$OUT/pmc_traffic.json" "$ROOT/profiles/r01_pmc_traffic.json"          # (the bench line quotes the latest traffic file)
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/stats.log" 2>&1
cp $(find "$OUT/stats" -name run_kernel_stats.csv) "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats"
cd "$ROOT"
timeout -k 10 300 python scripts/profile_layers.py > "$OUT/layers.txt" 2>&1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json"
