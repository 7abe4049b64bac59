cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s28
timeout 600 python -X faulthandler tools/s2s_shape_sweep.py 2>&1 | grep -v "amdgpu\|Extension modules" | tee gpurun_out/r6s28/sweep.txt
