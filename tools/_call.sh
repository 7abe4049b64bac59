cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6cs
timeout 1200 bash tools/ab_env.sh 4 SA_GRU_EXP=256 - 2>&1 | tee gpurun_out/r6cs/ab.txt
timeout 900 bash tools/gpu_run.sh r6cs "tests:baseline_configs or dropout or 16_byte"
