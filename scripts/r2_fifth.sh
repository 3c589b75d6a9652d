#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2e
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA > "$OUT/tests.log" 2>&1
echo "tests rc=$?" | tee -a "$OUT/tests.log"
grep -E "passed|failed" "$OUT/tests.log" | tail -3
grep -E "^FAILED|bf16x3: " "$OUT/tests.log" | head -30
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -c 2000 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
