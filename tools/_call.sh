cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s12
timeout 300 python tools/s2s_bwd_compare.py 2>&1 | tee gpurun_out/r6s12/cmp.log | grep -v amdgpu.ids 
timeout 900 bash tools/gpu_run.sh r6s12 "tests:seq2seq or s2s or Seq2Seq or config_4 or options or attention or decode or beam or greedy"
timeout 300 python tools/s2s_train_profile.py 20 > gpurun_out/r6s12/plain.log 2>&1; tail -1 gpurun_out/r6s12/plain.log
timeout 600 bash tools/gpu_run.sh r6s12 "profpy:tools/s2s_train_profile.py 10" | head -12 | cut -c1-150
