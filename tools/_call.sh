cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5s2j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_dropout.py -x -q 2>&1 | tail -3
export TRACE_MIN_US=0
bash tools/gpu_run.sh r5s2j "trace:tools/step_bench.py --no-prof" > /dev/null 2>&1
sed -n 1,8p $O/step_bench_timeline.txt | cut -c1-130; sed -n 22,45p $O/step_bench_timeline.txt | cut -c1-130
