#!/bin/bash
# A/B of environment switches on ONE box:  scripts/gpu.sh 1200 'bash scripts/r2_ab.sh "WSEG_PCM_STREAM=0" "WSEG_PCM_STREAM=1"'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-steps 0 $BENCH_ARGS > "$OUT/b.json" 2> "$OUT/b.err" || { echo "bench failed"; tail -3 "$OUT/b.err"; }
  echo "$cfg: $(python -c "import json;d=json.load(open('$OUT/b.json'));print(d['ms_per_step'], d['roofline']['frac'])")"
done
done
