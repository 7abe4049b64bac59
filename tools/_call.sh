cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_run.sh r6cj "tests:16_byte"
