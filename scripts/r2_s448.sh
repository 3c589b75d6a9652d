#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2s
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_loss.py -m gpu -q -k "full_resolution" -rA > "$OUT/tests.log" 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " "$OUT/tests.log" | cut -c1-400 | head -20
timeout -k 10 300 python scripts/measure_bf16_step.py bf16 > "$OUT/bf16_dev.json" 2> "$OUT/bf16_dev.err"; echo "measure rc=$?"
timeout -k 10 300 python scripts/measure_bf16_step.py bf16x3 > "$OUT/bf16x3_dev.json" 2> "$OUT/bf16x3_dev.err"; echo "measure rc=$?"
