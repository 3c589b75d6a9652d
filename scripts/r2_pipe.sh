#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/pipe
timeout -k 10 900 python scripts/bench_data_pipeline.py 1024 8,16 > gpurun_out/pipe/r02_data_pipeline.txt 2>&1; echo "pipeline rc=$?"; grep -v "Warn\|super\|amdgpu.ids" gpurun_out/pipe/r02_data_pipeline.txt
