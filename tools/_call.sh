cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s16
timeout 900 bash tools/gpu_run.sh r6s16 "tests:beam or seq2seq or s2s or decode"
timeout 600 bash tools/gpu_run.sh r6s16 "configs:M-S2S"
