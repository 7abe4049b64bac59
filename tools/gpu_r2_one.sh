#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/one; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_blocks.py -m gpu -x -q -k "fused_backward" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for one in 1 0; do
SA_GRU_BWD_ONE=$one timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one=$one', r['ms_per_step'], r['loss_rel_err'], r['persist_status'], r['roofline']['frac'], {k:round(v,3) for k,v in r['kernel_time_ms_per_step'].items()})"
done
