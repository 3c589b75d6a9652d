#!/bin/bash
# full verification + the measurements quoted in DESIGN.md (run through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2z
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA > "$OUT/tests.log" 2>&1
echo "tests rc=$?" | tee -a "$OUT/tests.log"
grep -E "passed|failed" "$OUT/tests.log" | tail -3
grep -E "^FAILED" "$OUT/tests.log" | head -30
WSEG_STREAMS=0 timeout -k 10 300 python -m pytest tests/test_gpu_loss.py -m gpu -q -k "step_matches or lookahead" > "$OUT/tests_nostreams.log" 2>&1; echo "single-stream tests rc=$?"; tail -1 "$OUT/tests_nostreams.log"
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > "$OUT/r02_bench_b16_448_bf16.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -c 2600 "$OUT/r02_bench_b16_448_bf16.json"
# the A/B switches of the scheduling features still work (fallback paths)
WSEG_BWD_PAIR=0 WSEG_PCM_STREAM=0 WSEG_PREFETCH=0 timeout -k 10 400 python -m pytest tests/test_gpu_loss.py -m gpu -q -k "step_matches or lookahead or full_step or packs_follow" > "$OUT/tests_switches_off.log" 2>&1; echo "switches-off tests rc=$?"; tail -1 "$OUT/tests_switches_off.log"
