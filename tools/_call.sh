cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6cw
SA_GRU_EXP=1 timeout 600 bash tools/gpu_run.sh r6cw "tests:planes"
timeout 900 bash tools/ab_env.sh 4 - SA_GRU_EXP=1 2>&1 | tee gpurun_out/r6cw/ab.txt
