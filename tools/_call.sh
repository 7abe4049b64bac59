# scratch: the command list of the current gpurun call (overwritten per call; see tools/README.md)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_run.sh r5fI tests smoke "bench:--steps 20 --warmup 5 --no-cpu-baseline --headline-only"
export TRACE_MIN_US=0
bash tools/gpu_run.sh r5fI "trace:tools/step_bench.py --no-prof --steps 4" > /dev/null 2>&1
grep "fold\|^step" gpurun_out/r5fI/step_bench_timeline.txt | cut -c1-110
