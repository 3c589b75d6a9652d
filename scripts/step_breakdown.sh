#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bd
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr" -o run -- python "$ROOT/bench.py" --steps 8 --warmup 3 --no-cpu-baseline --no-extras --parity-steps 0 > "$OUT/run.log" 2>&1 &&
python "$ROOT/scripts/step_breakdown.py" $(find "$OUT/tr" -name run_kernel_trace.csv) 4 > "$OUT/breakdown.txt" 2>&1 &&
python "$ROOT/scripts/step_timeline.py" $(find "$OUT/tr" -name run_kernel_trace.csv) > "$OUT/timeline.txt" 2>&1; echo "rc=$?"
rm -rf "$OUT/tr"
head -12 "$OUT/breakdown.txt"; grep "conv/wgrad launches" "$OUT/timeline.txt"
