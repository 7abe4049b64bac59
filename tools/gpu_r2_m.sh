#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_blocks.py -x -q -k "conv" 2>&1 | tail -15 )
( timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py tests/test_gpu_train_eval.py -x -q 2>&1 | tail -4 )
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['value'], r['kernel_time_ms_per_step'])"
timeout 300 python tools/bi_bench.py timit 2>&1 | grep -v amdgpu
