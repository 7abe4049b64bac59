#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export SA_GRU_SPIN_LIMIT=200000
for v in 0 1 2 4 6 3 5; do
  echo "== variant $v"; SA_GRU_BWD_VARIANT=$v timeout 120 python tools/gru_bwd_timing.py 2>&1 | grep -v amdgpu
done
