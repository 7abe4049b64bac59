# scratch: the command list of the current gpurun call (overwritten per call; see tools/README.md)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_run.sh r6final6 "bench:--steps 20 --warmup 5" "prof:--steps 20 --warmup 5 --no-cpu-baseline --headline-only"
export TRACE_MIN_US=0
bash tools/gpu_run.sh r6final6 "trace:tools/step_bench.py --no-prof" > /dev/null 2>&1
mv gpurun_out/r6final6/step_bench_timeline.txt gpurun_out/r6final6/train_step_timeline.txt
head -2 gpurun_out/r6final6/train_step_timeline.txt
