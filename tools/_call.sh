cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6final7
timeout 2400 bash tools/gpu_run.sh r6final7 tests smoke "bench:--steps 20 --warmup 5" "prof:--steps 20 --warmup 5 --no-cpu-baseline --headline-only" configs
