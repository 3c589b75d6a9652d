#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2f
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_augment.py -m gpu -q -x > "$OUT/aug_tests.log" 2>&1; echo "aug tests rc=$?"; tail -15 "$OUT/aug_tests.log" | cut -c1-300
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
timeout -k 10 700 python scripts/bench_data_pipeline.py 1024 8,16 > "$OUT/r02_data_pipeline.txt" 2>&1; echo "pipeline rc=$?"; grep -v "Warn\|super" "$OUT/r02_data_pipeline.txt"
