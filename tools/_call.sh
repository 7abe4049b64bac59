cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s32
for i in 15 16 17; do timeout 300 python -X faulthandler tools/ctc_shape_sweep.py $i 2>&1 | grep -v "amdgpu\|Extension modules" | head -14; done | tee gpurun_out/r6s32/sweep.txt
