cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s19
for f in 0 1 2 3 4; do echo "force nq $f"; SA_GRU_EXP=$((f * 65536)) timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu | cut -c1-100; done | tee gpurun_out/r6s19/conv_nq.txt
