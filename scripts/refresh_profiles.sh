#!/bin/bash
# Runs ON the GPU box (through gpurun): PMC traffic passes, rocprofv3 kernel stats, per-layer profile, NCE sweep, inference rates and
# the bench line of the current build; everything lands under gpurun_out/refresh/ (copy what should be judged into profiles/ afterwards).
#   scripts/gpu.sh 1200 'bash scripts/refresh_profiles.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/refresh
R=${WSEG_ROUND:-r03}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o run -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --parity-steps 0 > "$OUT/pmc_fetch.log" 2>&1 && echo "pmc fetch ok" &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o run -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --parity-steps 0 > "$OUT/pmc_write.log" 2>&1 && echo "pmc write ok" &&
python "$ROOT/scripts/summarize_pmc.py" $(find "$OUT/pmc_fetch" -name run_counter_collection.csv) $(find "$OUT/pmc_write" -name run_counter_collection.csv) "$OUT/${R}_pmc_traffic.json" > "$OUT/pmc_summary.txt" &&
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras --parity-steps 0 > "$OUT/stats.log" 2>&1 &&
cp $(find "$OUT/stats" -name run_kernel_stats.csv) "$OUT/${R}_bench_b16_448_bf16_kernel_stats.csv" && rm -rf "$OUT/stats" && echo "stats ok" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats3" -o run -- python "$ROOT/bench.py" --precision bf16x3 --steps 4 --warmup 2 --no-cpu-baseline --no-extras --parity-steps 0 > "$OUT/stats3.log" 2>&1 &&
cp $(find "$OUT/stats3" -name run_kernel_stats.csv) "$OUT/${R}_bench_b16_448_bf16x3_kernel_stats.csv" && rm -rf "$OUT/stats3" && echo "stats x3 ok" &&
cd "$ROOT" &&
timeout -k 10 300 python scripts/profile_layers.py bf16 > "$OUT/${R}_layers_b16_448_bf16.txt" 2>&1 && echo "layers ok" &&
timeout -k 10 300 python scripts/bench_nce_sweep.py > "$OUT/${R}_nce_similarity_sweep.txt" 2>&1 && echo "sweep ok" &&
(for p in bf16 bf16x3 fp32; do timeout -k 10 200 python scripts/bench_infer.py $p 16; done) > "$OUT/${R}_infer_375x500.txt" 2>&1 && echo "infer ok" &&
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > "$OUT/${R}_bench_b16_448_bf16.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -c 2500 "$OUT/${R}_bench_b16_448_bf16.json"; cat "$OUT/${R}_infer_375x500.txt" | grep inference
