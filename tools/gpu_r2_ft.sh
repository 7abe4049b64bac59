#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ft; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_blocks.py -m gpu -x -q -k "fused_backward" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 200 python tools/gru_bwd_timing.py 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['loss_rel_err'], r['persist_status'], {k:round(v,3) for k,v in r['kernel_time_ms_per_step'].items()})"
