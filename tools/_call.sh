cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s21
timeout 900 bash tools/gpu_run.sh r6s21 "tests:decode or infer or model or beam"
timeout 600 bash tools/gpu_run.sh r6s21 "configs:M-DEC"
timeout 200 python tools/infer_profile.py timit 1 2>&1 | tail -1
timeout 200 python tools/infer_profile.py slibri 1 2>&1 | tail -1
