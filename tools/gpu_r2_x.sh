#!/bin/bash
O=gpurun_out/r2x; mkdir -p $O
export TMPDIR=/tmp
python tools/ctc_time.py 2>&1 | grep -v amdgpu.ids
SA_CTC_PROB=0 python tools/ctc_time.py 2>&1 | grep -v amdgpu.ids
SA_CTC_PROB=3 python tools/ctc_time.py 2>&1 | grep -v amdgpu.ids
( cd /tmp && rm -rf /tmp/prof_x && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o p -- python $GRAFT_REPO_ROOT/tools/ctc_time.py 50 > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1 )
find /tmp/prof_x -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python tools/kstats.py $O/kernel_stats.csv 10
SA_CTC_DBG=1 python tools/ctc_clock_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
