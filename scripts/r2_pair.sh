#!/bin/bash
# scripts/gpu.sh 1200 'bash scripts/r2_pair.sh': the joint dgrad + wgrad grid (wseg_conv_bwd_pair) — its tests, then the same-box A/B against two launches
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pair
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "pair" > "$OUT/conv.log" 2>&1; echo "pair tests rc=$?"; tail -2 "$OUT/conv.log" | cut -c1-300
for rep in 1 2; do
for cfg in "WSEG_BWD_PAIR=0" "WSEG_BWD_PAIR=1"; do
  env $cfg timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-steps 0 > "$OUT/b.json" 2> "$OUT/b.err" || { echo "bench failed"; tail -3 "$OUT/b.err"; }
  echo "$cfg: $(python -c "import json;d=json.load(open('$OUT/b.json'));print(d['ms_per_step'], d['roofline']['frac'])")"
done
done
