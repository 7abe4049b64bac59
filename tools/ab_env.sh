#!/bin/bash
# tools/ab_env.sh ROUNDS "ENV1" "ENV2" ... -- bench.py back to back on ONE box under each environment string
# ("VAR=a VAR2=b", "-" = none), ROUNDS times round robin; prints ms per step and the two stack times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=$1; shift
for i in $(seq $N); do
  for e in "$@"; do
    [ "$e" = "-" ] && ee="" || ee="$e"
    r=$(env $ee timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --headline-only 2>/dev/null | tail -1)
    echo "[$e] $(python -c "import json,sys; r=json.loads(sys.argv[1]); k=r.get('kernel_time_ms_per_step',{}); print(r['ms_per_step'], k.get('gru_fwd_stack'), k.get('gru_bwd_stack'), r['roofline'].get('kernel_us'), (r['roofline_other'].get('gru_fwd_step_kernel') or {}).get('kernel_us'))" "$r")"
  done
done
