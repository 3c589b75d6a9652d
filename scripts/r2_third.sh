#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2c
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA > "$OUT/tests.log" 2>&1
echo "tests rc=$?" | tee -a "$OUT/tests.log"
grep -E "passed|failed" "$OUT/tests.log" | tail -3
grep -E "^FAILED|mismatch fraction|CAM mIoU|edge fixture" "$OUT/tests.log" | head -30
timeout -k 10 300 python scripts/bench_nce_sweep.py > "$OUT/nce_sweep.txt" 2>&1
echo "sweep rc=$?"; cat "$OUT/nce_sweep.txt"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -c 2000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
