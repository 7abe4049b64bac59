cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TRACE_MIN_US=0
for r in 1 2; do bash tools/gpu_run.sh r6cv "trace:tools/step_bench.py --no-prof" > /dev/null 2>&1
head -1 gpurun_out/r6cv/step_bench_timeline.txt; grep "pk_pack" gpurun_out/r6cv/step_bench_timeline.txt | cut -c1-100; done
