cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6bx
cp speech_amd/libspeech_amd.so tools/_ab_libs/new.so
for r in 1 2 3; do for v in old new; do
  cp tools/_ab_libs/$v.so speech_amd/libspeech_amd.so; touch speech_amd/libspeech_amd.so
  echo "== $v"; bash tools/ab_env.sh 1 - 2>&1
done; done | tee gpurun_out/r6bx/ab.txt
cp tools/_ab_libs/new.so speech_amd/libspeech_amd.so; touch speech_amd/libspeech_amd.so
python tools/gru_bwd_timing.py 2>&1 | grep "all blocks"
bash tools/gpu_run.sh r6bx "tests:baseline_configs or fused or stack or dropout or health or fault"
