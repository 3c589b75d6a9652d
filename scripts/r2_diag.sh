#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 300 python scripts/diag_s448.py fp32 2>&1 | grep -v amdgpu.ids
