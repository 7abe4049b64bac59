cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s29
timeout 900 bash tools/gpu_run.sh r6s29 "tests:forty_random"
grep -n "^E  \|Error" gpurun_out/r6s29/pytest_forty_random.log | head -10
