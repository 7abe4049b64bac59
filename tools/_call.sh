cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_run.sh r5fE tests smoke "bench:--steps 20 --warmup 5"
export TRACE_MIN_US=0
bash tools/gpu_run.sh r5fE "trace:tools/step_bench.py --no-prof" > /dev/null 2>&1
head -40 gpurun_out/r5fE/step_bench_timeline.txt | cut -c1-120
