cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5c24; mkdir -p $O
timeout 600 python tools/ctc_flags_probe.py 1024 4096 32 2>&1 | grep -v amdgpu | tee $O/flags.log
timeout 300 python -m pytest tests/test_gpu_ctc.py -x -q 2>&1 | tail -2
