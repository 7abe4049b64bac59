cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6final10
timeout 2400 bash tools/gpu_run.sh r6final10 tests smoke "bench:--steps 20 --warmup 5"
