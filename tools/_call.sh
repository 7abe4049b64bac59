cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6cx
SA_GRU_EXP=1 timeout 600 bash tools/gpu_run.sh r6cx "tests:planes or baseline_configs"
timeout 900 bash tools/ab_env.sh 4 - SA_GRU_EXP=1 2>&1 | tee gpurun_out/r6cx/ab.txt
