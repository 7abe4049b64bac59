#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for v in 0 1; do
  echo "== variant $v"; SA_GRU_BWD_VARIANT=$v timeout 120 python tools/gru_bwd_timing.py 2>&1 | grep -v amdgpu
done
( timeout 900 python -m pytest tests/test_gpu_blocks.py -x -q -k "persist or fused or xcd or gru" 2>&1 | tail -3 )
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['avg_launch_us'], r['persist_status'], r['kernel_time_ms_per_step'])"
