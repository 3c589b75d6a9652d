#!/bin/bash
# same-box A/B of whole-step time (runs ON the GPU box): scripts/ab_bench.sh OUTDIR "NAME=ENV..." ...   e.g.
#   scripts/gpu.sh 900 'bash scripts/ab_bench.sh gpurun_out/ab "head=WSEG_LIB=$PWD/wseg_amd/libwseg_hip_head.so" "new=WSEG_CONV_LOOP=1" "head=..." "new=..."'
OUT=$1; shift; mkdir -p "$OUT"
i=0
for spec in "$@"; do
  i=$((i+1)); name=${spec%%=*}; envs=${spec#*=}
  env $envs timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --parity-steps 0 > "$OUT/$i.$name.json" 2> "$OUT/$i.$name.err" || { echo "$name failed"; tail -3 "$OUT/$i.$name.err"; exit 1; }
  python - "$OUT/$i.$name.json" "$name" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:12s} {d['ms_per_step']:7.2f} ms/step  conv frac {d['roofline']['frac']:.4f}  whole {d['roofline']['whole_step_frac']:.4f}  avg launch {d['roofline']['avg_launch_ms']:.4f} ms", flush=True)
P
done
