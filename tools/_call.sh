cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s15
timeout 600 bash tools/gpu_run.sh r6s15 "profpy:tools/timit_profile.py" > gpurun_out/r6s15/stage.log 2>&1
grep -E "conv_|Name" gpurun_out/r6s15/timit_profile_kernel_stats.csv | cut -c1-200
