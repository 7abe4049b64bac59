#!/bin/bash
# probability-domain CTC chain: parity in all three modes + timing
O=gpurun_out/r2w; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ctc.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'ctc_loss_step_ms', d['ctc_loss_step_ms'], 'ctc in-step', d['kernel_time_ms_per_step'].get('ctc_loss'))" | tee $O/bench.log
SA_CTC_PROB=0 timeout 300 python bench.py --no-cpu-baseline --steps 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LOG: ms_per_step', d['ms_per_step'], 'ctc_loss_step_ms', d['ctc_loss_step_ms'], 'ctc in-step', d['kernel_time_ms_per_step'].get('ctc_loss'))" | tee -a $O/bench.log
