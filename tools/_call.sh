cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s24
for i in 0 1 2 3 4 5 6 7 8; do timeout 300 python tools/s2s_shape_sweep.py $i 2>&1 | grep -v "amdgpu\|^  \.\." ; done | tee gpurun_out/r6s24/sweep.txt
timeout 900 bash tools/gpu_run.sh r6s24 "tests:gru_stack_matches_oracle"
