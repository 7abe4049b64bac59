cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s7
timeout 300 python tools/s2s_bwd_compare.py 2>&1 | tee gpurun_out/r6s7/cmp.log | grep -v amdgpu.ids
timeout 900 bash tools/gpu_run.sh r6s7 "tests:seq2seq or s2s or Seq2Seq or config_4 or options"
timeout 300 python tools/s2s_train_profile.py 20 > gpurun_out/r6s7/plain.log 2>&1; tail -1 gpurun_out/r6s7/plain.log
timeout 600 bash tools/gpu_run.sh r6s7 "profpy:tools/s2s_train_profile.py 10" | head -16 | cut -c1-150
for c in 1 2 3 4; do echo "chains $c"; SA_S2S_CHAINS=$c timeout 300 python tools/s2s_train_profile.py 20 2>&1 | tail -1; done
