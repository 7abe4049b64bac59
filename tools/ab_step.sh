#!/bin/bash
# tools/ab_step.sh ROUNDS "STEP_BENCH ARGS" "ENV1" "ENV2" ... -- tools/step_bench.py back to back on ONE box under each
# environment string ("-" = none), ROUNDS times round robin.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=$1; A=$2; shift; shift
for i in $(seq $N); do
  for e in "$@"; do
    [ "$e" = "-" ] && ee="" || ee="$e"
    echo "[$e] $(env $ee timeout 300 python tools/step_bench.py $A --no-prof 2>/dev/null | tail -1 | cut -c1-160)"
  done
done
