cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s26
timeout 900 bash tools/gpu_run.sh r6s26 "tests:edge_shapes"
tail -15 gpurun_out/r6s26/pytest_edge_shapes.log
