#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/mfma
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p" -o run -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --parity-steps 0 > "$OUT/run.log" 2>&1; echo "rc=$?"
CC=$(find "$OUT/p" -name run_counter_collection.csv); KT=$(find "$OUT/p" -name run_kernel_trace.csv)
head -2 "$CC" | cut -c1-400
python "$ROOT/scripts/summarize_mfma.py" "$CC" "$KT" "$OUT/${WSEG_ROUND:-r03}_mfma_utilisation.json"; echo "sum rc=$?"
rm -rf "$OUT/p"
