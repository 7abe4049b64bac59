cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s30
timeout 900 bash tools/gpu_run.sh r6s30 "tests:thirty_random"
grep -n "^E  \|Error" gpurun_out/r6s30/pytest_thirty_random.log | head -10
