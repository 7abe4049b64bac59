#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -4 )
echo "== overlap on (XCD-filtered side GEMMs)"; timeout 300 python tools/bi_bench.py 2>&1 | grep -v amdgpu
echo "== overlap off"; SA_GRU_OVERLAP=0 timeout 300 python tools/bi_bench.py 2>&1 | grep -v amdgpu
