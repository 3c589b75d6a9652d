#!/bin/bash
# round-2 first GPU pass: tests (bars not yet set for the bf16 tests), bf16 deviation measurement, bench line with parity + CPU legs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2a
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -rA --deselect tests/test_gpu_loss.py::test_bf16_step_against_reference_fixture --deselect tests/test_gpu_loss.py::test_bf16_three_steps_against_reference_fixture > "$OUT/tests.log" 2>&1
echo "tests rc=$?" | tee -a "$OUT/tests.log"
grep -E "passed|failed" "$OUT/tests.log" | tail -3
timeout -k 10 300 python scripts/measure_bf16_step.py bf16 > "$OUT/bf16_dev.json" 2> "$OUT/bf16_dev.err"
echo "measure rc=$?"
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
tail -c 3000 "$OUT/bench.json"
