#!/bin/bash
# HBM traffic of the CTC-loss kernels at M-CTC B = 32 and B = 4096 (tools/ctc_pmc_run.py): rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE in separate passes, --kernel-trace only -> gpurun_out/TAG/ctc_traffic.json (tools/pmc_traffic.py:
# bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of MI355X_MICROARCH.md).  usage: tools/pmc_ctc_traffic.sh TAG
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-ctcpmc}; O=$R/gpurun_out/$TAG; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ct_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/ct_$c -o p -- python $R/tools/ctc_pmc_run.py > $O/ctc_pmc_$c.log 2>&1
done
python $R/tools/pmc_traffic.py /tmp/ct_FETCH_SIZE /tmp/ct_WRITE_SIZE ctc_ > $O/ctc_traffic.json
grep -E '"ctc_|bytes_per_launch' $O/ctc_traffic.json | paste - - | cut -c1-160
