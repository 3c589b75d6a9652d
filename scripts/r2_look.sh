#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/look
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_loss.py tests/test_gpu_ddp_equivalence.py tests/test_gpu_infer_cli.py tests/test_gpu_net.py tests/test_gpu_contract.py -m gpu -q -x > "$OUT/tests.log" 2>&1; echo "tests rc=$?"; tail -5 "$OUT/tests.log" | cut -c1-300
for look in "" "--no-lookahead" ""; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-steps 0 $look > "$OUT/b.json" 2> "$OUT/b.err" || { echo "bench failed"; tail -3 "$OUT/b.err"; }
  echo "bench $look: $(python -c "import json;d=json.load(open('$OUT/b.json'));print(d['ms_per_step'], d['roofline']['frac'])")"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr" -o run -- python "$ROOT/bench.py" --steps 8 --warmup 3 --no-cpu-baseline --parity-steps 0 > "$OUT/run.log" 2>&1 &&
python "$ROOT/scripts/step_breakdown.py" $(find "$OUT/tr" -name run_kernel_trace.csv) 4 > "$OUT/breakdown.txt" 2>&1 &&
python "$ROOT/scripts/step_timeline.py" $(find "$OUT/tr" -name run_kernel_trace.csv) > "$OUT/timeline.txt" 2>&1; echo "rc=$?"
rm -rf "$OUT/tr"
head -4 "$OUT/breakdown.txt"; grep "up_stats_partial" "$OUT/breakdown.txt"
