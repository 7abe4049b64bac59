#!/bin/bash
# Round-2 evidence: headline bench JSON, rocprof kernel stats, PMC HBM traffic (separate FETCH / WRITE passes), all configs.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r02prof; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_blocks.py -x -q -k conv 2>&1 | tail -2 )
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
( cd /tmp && rm -rf /tmp/p_stats && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/$O/rocprof_stats.log 2>&1 )
find /tmp/p_stats -name "*kernel_stats.csv" -exec cp {} $O/train_step_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/p_$c && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/rocprof_$c.log 2>&1 )
done
python tools/pmc_traffic.py /tmp/p_FETCH_SIZE /tmp/p_WRITE_SIZE gru_ gemm_f32 conv1 > $O/pmc_hbm_traffic.json 2> $O/pmc_traffic.err
timeout 900 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err
timeout 300 python tools/bi_bench.py > $O/bi_bench.txt 2>&1
python - <<PY
import json
r = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["kernel_time_ms_per_step"])
print(open("$O/pmc_hbm_traffic.json").read()[:1500])
PY
