#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2d
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA > "$OUT/tests.log" 2>&1
echo "tests rc=$?" | tee -a "$OUT/tests.log"
grep -E "passed|failed" "$OUT/tests.log" | tail -3
grep -E "^FAILED|mismatch fraction|CAM mIoU|keys differ|largest top-1" "$OUT/tests.log" | head -30
timeout -k 10 300 python scripts/measure_bf16_step.py bf16x3 > "$OUT/bf16x3_dev.json" 2> "$OUT/bf16x3_dev.err"; echo "measure rc=$?"
WSEG_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --parity-steps 0 > "$OUT/bench_2rank_gloo.json" 2> "$OUT/bench_2rank_gloo.err"
echo "2-rank rc=$?"; tail -c 600 "$OUT/bench_2rank_gloo.json"; tail -3 "$OUT/bench_2rank_gloo.err"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_x3" -o run -- python "$ROOT/bench.py" --precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --parity-steps 0 > "$OUT/stats_x3.log" 2>&1
echo "rocprof rc=$?"
cp $(find "$OUT/stats_x3" -name run_kernel_stats.csv) "$OUT/x3_kernel_stats.csv" && rm -rf "$OUT/stats_x3"
head -12 "$OUT/x3_kernel_stats.csv" | cut -c1-200
