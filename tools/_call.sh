cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5s2q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_health_dist.py -x -q -k "sgd or clip or train or step or health" 2>&1 | tail -2
export TRACE_MIN_US=0
for d in 0 1; do
  bash tools/gpu_run.sh r5s2q "trace:tools/step_bench.py --no-prof --steps 6" > /dev/null 2>&1
  echo "$(grep 'sumsq\|clip_sgd' $O/step_bench_timeline.txt | cut -c1-90 | tr '\n' ' ') $(head -1 $O/step_bench_timeline.txt)"
done
cp $O/step_bench_timeline.txt $O/train_step_timeline.txt
