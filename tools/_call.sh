cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5s2g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_blocks.py -x -q -k "thin or gemm" 2>&1 | tail -3
export TRACE_MIN_US=0
bash tools/gpu_run.sh r5s2g "trace:tools/step_bench.py --no-prof" > /dev/null 2>&1
sed -n 1,30p $O/step_bench_timeline.txt | cut -c1-130
