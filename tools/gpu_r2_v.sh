#!/bin/bash
# kernel stats of the TIMIT-shaped train step
O=gpurun_out/r2v; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_v && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -o p -- python $GRAFT_REPO_ROOT/tools/bi_bench.py timit > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1 )
find /tmp/prof_v -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python tools/kstats.py $O/kernel_stats.csv 30
