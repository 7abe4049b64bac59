cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6cy
SA_GRU_EXP=256 timeout 900 bash tools/gpu_run.sh r6cy "tests:baseline_configs or shared_packed or model"
timeout 1200 bash tools/ab_env.sh 5 - SA_GRU_EXP=256 2>&1 | tee gpurun_out/r6cy/ab.txt
