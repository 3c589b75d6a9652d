#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2n
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_loss.py -m gpu -q -x -k "fused_nce or step_matches" > "$OUT/tests.log" 2>&1; echo "tests rc=$?"; tail -5 "$OUT/tests.log" | cut -c1-300
timeout -k 10 300 python scripts/bench_nce_sweep.py > "$OUT/sweep.txt" 2>&1; echo "sweep rc=$?"; cat "$OUT/sweep.txt"
