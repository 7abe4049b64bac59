cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6final8
timeout 900 bash tools/gpu_run.sh r6final8 "bench:--steps 20 --warmup 5" "tests:health or bench"
