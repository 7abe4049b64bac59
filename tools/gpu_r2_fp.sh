#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/fp; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_blocks.py -m gpu -x -q -k "xcd_local or persistent or bidirectional" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/bi_fwd_time.py 2>&1 | tail -1
timeout 300 python tools/uni_bench.py 2>&1 | tail -4
