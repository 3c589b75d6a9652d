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
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout -k 10 300 python scripts/bench_nce_sweep.py > "$OUT/r02_nce_similarity_sweep.txt" 2>&1; echo "sweep rc=$?"; cat "$OUT/r02_nce_similarity_sweep.txt"
timeout -k 10 700 python scripts/bench_data_pipeline.py 1024 8,16 > "$OUT/r02_data_pipeline.txt" 2>&1; echo "pipeline rc=$?"; grep -v "Warn\|super\|amdgpu.ids" "$OUT/r02_data_pipeline.txt"
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > "$OUT/r02_bench_b16_448_bf16.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -c 2600 "$OUT/r02_bench_b16_448_bf16.json"
