#!/bin/bash
O=gpurun_out/r2y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_blocks.py -q -x -m gpu -k "gemm or gru or stack" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for f in 1 0; do
SA_GEMM_FOLD=$f timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FOLD=$f ms_per_step', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_time_ms_per_step'].items()})"
done
timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | sed -n 4,9p
