#!/bin/bash
# second half of the round-2 evidence: PMC HBM traffic of the headline step (separate FETCH / WRITE passes), all configs
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r02prof; mkdir -p $O
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/p_$c && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/rocprof_$c.log 2>&1 )
done
python tools/pmc_traffic.py /tmp/p_FETCH_SIZE /tmp/p_WRITE_SIZE gru_ gemm_f32 conv > $O/pmc_hbm_traffic.json 2> $O/pmc_traffic.err
head -c 1800 $O/pmc_hbm_traffic.json
( time timeout 500 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err ) 2>&1 | tail -3
python - <<PY 2>/dev/null
import json
for r in json.load(open("$O/configs.json")) if open("$O/configs.json").read().strip().startswith("[") else []:
    print(r)
PY
tail -c 1500 $O/configs.json
