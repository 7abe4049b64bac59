cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s13
timeout 1500 bash tools/gpu_run.sh r6s13 "tests:seq2seq or s2s or Seq2Seq or config or transducer or rnnt or model or train or flat or dist or linear"
timeout 300 python tools/s2s_train_profile.py 20 > gpurun_out/r6s13/plain.log 2>&1; tail -1 gpurun_out/r6s13/plain.log
timeout 600 bash tools/gpu_run.sh r6s13 "configs:M-S2S,M-RNNT"
