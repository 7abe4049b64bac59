# scratch: the command list of the current gpurun call (overwritten per call; see tools/README.md)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/final_pass.sh r6final3 2>&1 | tail -60
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_run.sh r6final3 "bench:--steps 20 --warmup 5" "prof:--steps 20 --warmup 5 --no-cpu-baseline --headline-only"
