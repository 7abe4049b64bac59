#!/bin/bash
# PMC HBM traffic of the CTC kernels at B=32 and B=4096 (separate FETCH_SIZE / WRITE_SIZE passes, kernel trace only)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r02pmc; mkdir -p $O
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/pc_$c && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pc_$c -o p -- python $R/tools/ctc_pmc_run.py > $R/$O/rocprof_$c.log 2>&1 )
done
python tools/pmc_traffic.py /tmp/pc_FETCH_SIZE /tmp/pc_WRITE_SIZE ctc_ > $O/pmc_ctc.json 2> $O/pmc_ctc.err
cat $O/pmc_ctc.json
