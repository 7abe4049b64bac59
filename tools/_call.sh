cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6cc
timeout 900 bash tools/ab_env.sh 3 - SA_GRU_EXP=128 SA_GRU_EXP=256 2>&1 | tee gpurun_out/r6cc/ab.txt
