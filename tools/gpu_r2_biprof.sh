#!/bin/bash
# rocprof kernel stats of the bidirectional S-LIBRI train step (tools/bi_fwd_time.py: 3 + 10 forwards, 3 + 10 train steps)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/biprof; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_bi && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bi -o p -- python $R/tools/bi_fwd_time.py > $R/$O/rocprof.log 2>&1 )
find /tmp/prof_bi -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
tail -1 $O/rocprof.log; head -12 $O/kernel_stats.csv | cut -c1-150
