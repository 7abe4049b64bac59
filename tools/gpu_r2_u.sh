#!/bin/bash
O=gpurun_out/r2u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_baseline_configs.py tests/test_gpu_train_eval.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench.log
timeout 300 python tools/bi_bench.py timit 2>&1 | tail -1 | tee $O/timit.log
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee $O/bench.log
