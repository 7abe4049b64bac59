cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5s2p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_blocks.py -x -q -k "conv" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py tests/test_gpu_seq2seq.py -x -q 2>&1 | tail -2
export TRACE_MIN_US=0
for d in 0 0; do
  bash tools/gpu_run.sh r5s2p "trace:tools/step_bench.py --no-prof --steps 4" > /dev/null 2>&1
  echo "$(grep conv_dw2_kernel $O/step_bench_timeline.txt | cut -c1-100) $(head -1 $O/step_bench_timeline.txt)"
done
bash tools/gpu_run.sh r5s2p "trace:tools/step_bench.py --case timit --no-prof --steps 4" > /dev/null 2>&1
grep "conv_dw2\|^step" $O/step_bench_timeline.txt | cut -c1-110
