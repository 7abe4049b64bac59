cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s8
timeout 400 bash tools/gpu_run.sh r6s8 "profpy:tools/rnnt_profile.py" | head -30 | cut -c1-170
TRACE_MIN_US=0 timeout 400 bash tools/gpu_run.sh r6s8 "trace:tools/rnnt_profile.py" > gpurun_out/r6s8/trace_stage.log 2>&1
timeout 600 bash tools/gpu_run.sh r6s8 "configs:M-S2S"
