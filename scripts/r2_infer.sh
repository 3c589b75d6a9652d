#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/inf

timeout -k 10 900 python scripts/bench_infer_cli.py 1449 bf16 8 > gpurun_out/inf/r02_infer_cli_1449.txt 2>&1; echo "rc=$?"; grep -v "Warn\|warn\|amdgpu.ids" gpurun_out/inf/r02_infer_cli_1449.txt | tail -8
