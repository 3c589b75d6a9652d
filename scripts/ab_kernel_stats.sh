#!/bin/bash
# kernel-stats A/B of two builds on one box (runs ON the GPU box): bash scripts/ab_kernel_stats.sh OUT "name=ENV" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/$1; shift; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  export $envs
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o run -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras --parity-steps 0 > "$OUT/$name.log" 2>&1 || { echo "$name failed"; tail -3 "$OUT/$name.log"; }
  unset ${envs%%=*}
  cp $(find "$OUT/$name" -name run_kernel_stats.csv) "$OUT/$name.kernel_stats.csv"; rm -rf "$OUT/$name"
  echo "== $name"; head -12 "$OUT/$name.kernel_stats.csv" | cut -c1-200
done
