#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pair
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x > "$OUT/conv.log" 2>&1; echo "conv tests rc=$?"; tail -3 "$OUT/conv.log" | cut -c1-300
timeout -k 10 900 python -m pytest tests/test_gpu_loss.py tests/test_gpu_net.py tests/test_gpu_ddp_equivalence.py -m gpu -q -x > "$OUT/tests.log" 2>&1; echo "tests rc=$?"; tail -3 "$OUT/tests.log" | cut -c1-300
for rep in 1 2; do
for cfg in "WSEG_BWD_PAIR=0" "WSEG_BWD_PAIR=1"; do
  env $cfg timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-steps 0 > "$OUT/b.json" 2> "$OUT/b.err" || { echo "bench failed"; tail -3 "$OUT/b.err"; }
  echo "$cfg: $(python -c "import json;d=json.load(open('$OUT/b.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['launches_per_step'])")"
done
done
