cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s27
for i in 0 1 2 3 4 5 6 7; do timeout 300 python -X faulthandler tools/rnnt_shape_sweep.py $i 2>&1 | grep -v "amdgpu\|all shapes agree\|Extension modules\|UserWarning\|super().__init__" | head -16; done | tee gpurun_out/r6s27/sweep.txt
