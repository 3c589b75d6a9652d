#!/bin/bash
# board power / clocks while the bench runs (ON the GPU box): is the step running at the power cap?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
python "$ROOT/bench.py" --steps 600 --warmup 5 --no-cpu-baseline --no-extras --parity-steps 0 > /tmp/bench_power.json 2> /tmp/bench_power.err &
BP=$!
for i in $(seq 1 120); do
  w=$(rocm-smi --showpower 2>/dev/null | grep -i "Power (W)" | sed 's/.*: //' | cut -d. -f1)
  if [ -n "$w" ] && [ "$w" -gt 400 ]; then echo "t=$i power ${w} W  $(rocm-smi --showclocks 2>/dev/null | grep -i sclk | sed 's/.*(//; s/).*//')"; fi
  kill -0 $BP 2>/dev/null || break
  sleep 1
done
wait $BP; python -c "
import json; d=json.loads(open('/tmp/bench_power.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], 'ms/step')"
