#!/bin/bash
O=gpurun_out/r2z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_baseline_configs.py -q -x -m gpu -k "bi or stack or gru or timit or TIMIT or config" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for c in 1 2 4 6 8; do
SA_GRU_FWD_CHUNKS=$c timeout 300 python tools/bi_bench.py 2>&1 | grep -v amdgpu | cut -c1-230 | sed "s/^/chunks=$c /"
done
