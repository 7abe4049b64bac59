#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "0 0" "0 1" "1 0" "1 1"; do
  set -- $cfg
  env SA_GRU_FUSE_DX=$1 $( [ "$2" = 1 ] && echo SA_GRU_DBG_HOT=1 ) timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_dx=$1 hot=$2', r['ms_per_step'], r['persist_status'], {k:round(v,3) for k,v in r['kernel_time_ms_per_step'].items()})"
done
