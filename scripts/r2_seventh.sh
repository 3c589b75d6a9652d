#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2g
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 200 python scripts/prof_augment.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -3
timeout -k 10 700 python scripts/bench_data_pipeline.py 1024 8,16 > "$OUT/r02_data_pipeline.txt" 2>&1; echo "pipeline rc=$?"; grep -v "Warn\|super\|amdgpu.ids" "$OUT/r02_data_pipeline.txt"
