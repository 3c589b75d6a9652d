#!/bin/bash
# counter passes over one conv layer (runs ON the GPU box): scripts/gpu.sh 600 'bash scripts/pmc_conv_layer.sh 512 conv_igemm256'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-512}; K=${2:-conv_igemm256}; HINT=${3:-0}
OUT=$ROOT/gpurun_out/pmc_$W; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o run -- python "$ROOT/scripts/conv_one.py" $W 10 $HINT > "$OUT/run$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/run$i.log"; }
done
python "$ROOT/scripts/summarize_counters.py" "$K" $(find "$OUT" -name run_counter_collection.csv) | tee "$OUT/summary.txt"
cat "$OUT"/run1.log | tail -2
rm -rf "$OUT"/p*
