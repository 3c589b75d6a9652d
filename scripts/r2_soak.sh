#!/bin/bash
# scripts/gpu.sh 1200 'bash scripts/r2_soak.sh': allocator numbers of the fused step per mode (fresh processes), then the 150-step trajectories
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/soak
(timeout -k 10 300 python scripts/soak_lookahead.py 150 mem 0 && timeout -k 10 300 python scripts/soak_lookahead.py 150 mem 1 && timeout -k 10 300 python scripts/soak_lookahead.py 400 mem 1 && timeout -k 10 600 python scripts/soak_lookahead.py 150) > gpurun_out/soak/soak.txt 2>&1; echo "soak rc=$?"; grep -v "Warn\|warn\|amdgpu.ids" gpurun_out/soak/soak.txt | tail -8
