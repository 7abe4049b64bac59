#!/bin/bash
# tools/ab_bench.sh VAR A B [ROUNDS] -- bench.py back to back on ONE box with VAR=A and VAR=B alternating (boxes differ by
# ~1 % from each other, so a switch worth less than that can only be judged inside one call).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for v in "$A" "$B"; do
    r=$(env $VAR=$v timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$VAR=$v $(python -c "import json,sys; r=json.loads(sys.argv[1]); print(r['ms_per_step'], r.get('kernel_time_ms_per_step',{}).get('gru_bwd_stack'))" "$r")"
  done
done
