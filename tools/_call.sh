cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6final11
timeout 2400 bash tools/gpu_run.sh r6final11 tests smoke
