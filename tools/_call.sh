cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s35
timeout 900 bash tools/gpu_run.sh r6s35 "tests:dropout"
timeout 600 bash tools/gpu_run.sh r6s35 "profpy:tools/step_bench.py --dropout 0.2 --no-prof" > /dev/null 2>&1
grep "gru_fwd_planes\|gru_bwd_fused16" gpurun_out/r6s35/step_bench_kernel_stats.csv | cut -c1-160
timeout 300 python tools/step_bench.py --dropout 0.2 --no-prof 2>&1 | tail -2
timeout 300 python tools/step_bench.py --no-prof 2>&1 | tail -2
