cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6ct
timeout 1200 bash tools/ab_env.sh 4 - SA_GRU_EXP=256 SA_GRU_EXP=512 2>&1 | tee gpurun_out/r6ct/ab.txt
