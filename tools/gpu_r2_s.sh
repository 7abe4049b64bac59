#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
echo "== overlap on"; SA_GRU_OVERLAP=1 timeout 300 python tools/bi_bench.py 2>&1 | grep -v amdgpu
echo "== overlap off"; timeout 300 python tools/bi_bench.py 2>&1 | grep -v amdgpu
( timeout 600 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py -x -q -k "bidirectional or bi2 or conv or matches" 2>&1 | tail -2 )
