# scratch: the command list of the current gpurun call (overwritten per call; see tools/README.md)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_run.sh r6final7 tests smoke
