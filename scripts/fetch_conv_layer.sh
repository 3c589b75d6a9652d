#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one conv layer (ON the GPU box): bash scripts/fetch_conv_layer.sh 512
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-512}; OUT=$ROOT/gpurun_out/fetch_$W; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/p1" -o run -- python "$ROOT/scripts/conv_one.py" $W 10 > "$OUT/run1.log" 2>&1
python "$ROOT/scripts/summarize_counters.py" conv_igemm256 $(find "$OUT" -name run_counter_collection.csv); tail -1 "$OUT/run1.log"
rm -rf "$OUT"/p*
