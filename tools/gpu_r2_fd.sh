#!/bin/bash
# fused backward input gradient: tests, then A/B on the headline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/fd; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_blocks.py -m gpu -x -q -k "fused_backward or xcd_local or persistent" ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for fd in 0 1 0 1; do
  SA_GRU_FUSE_DX=$fd timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_dx=$fd', r['value'], r['ms_per_step'], r['loss_rel_err'], r['persist_status'], {k:round(v,3) for k,v in r['kernel_time_ms_per_step'].items()} if isinstance(r.get('kernel_time_ms_per_step'),dict) else '')"
done
