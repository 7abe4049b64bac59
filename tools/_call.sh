cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s17
timeout 300 python tools/s2s_bwd_compare.py 2>&1 | tee gpurun_out/r6s17/cmp.log | grep -v amdgpu.ids | grep "dim\|loss\|dec_rnn.weight_ih\|fc.fc.weight"
timeout 900 bash tools/gpu_run.sh r6s17 "tests:seq2seq or s2s or Seq2Seq or config_4 or options or attention or decode or beam or greedy"
for k in 7 3; do echo "kernels $k"; SA_S2S_KERNELS=$k timeout 300 python tools/s2s_train_profile.py 20 2>&1 | tail -1; done
timeout 600 bash tools/gpu_run.sh r6s17 "profpy:tools/s2s_train_profile.py 10" | grep "attention\|skinny" | cut -c1-150
