#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2h
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA > "$OUT/tests.log" 2>&1
echo "tests rc=$?" | tee -a "$OUT/tests.log"
grep -E "passed|failed" "$OUT/tests.log" | tail -3
grep -E "^FAILED" "$OUT/tests.log" | head -30
timeout -k 10 300 python scripts/bench_nce_sweep.py > "$OUT/r02_nce_similarity_sweep.txt" 2>&1; echo "sweep rc=$?"; cat "$OUT/r02_nce_similarity_sweep.txt"
timeout -k 10 300 python scripts/measure_bf16_step.py bf16 > "$OUT/bf16_dev.json" 2> "$OUT/bf16_dev.err"; echo "measure rc=$?"
timeout -k 10 300 python scripts/measure_bf16_step.py bf16x3 > "$OUT/bf16x3_dev.json" 2> "$OUT/bf16x3_dev.err"; echo "measure rc=$?"
