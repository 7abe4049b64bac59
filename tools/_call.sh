cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s33
timeout 900 bash tools/gpu_run.sh r6s33 "tests:vector_and_block_boundaries"
grep -n "^E  \|Error" gpurun_out/r6s33/pytest_vector_and_block_boundaries.log | head -10
