#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/look
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
for n in 0 128 64; do
  WSEG_PREFIX_TILE=$n timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-steps 0 > "$OUT/b.json" 2> "$OUT/b.err" || { echo "bench failed"; tail -3 "$OUT/b.err"; }
  echo "prefix tile $n: $(python -c "import json;d=json.load(open('$OUT/b.json'));print(d['ms_per_step'], d['roofline']['frac'])")"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-steps 0 --no-lookahead > "$OUT/b.json" 2> "$OUT/b.err"; echo "no lookahead: $(python -c "import json;d=json.load(open('$OUT/b.json'));print(d['ms_per_step'], d['roofline']['frac'])")"
