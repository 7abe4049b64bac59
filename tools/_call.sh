cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6ca
for r in 1 2 3; do for v in old new; do
  cp tools/_ab_libs/$v.so speech_amd/libspeech_amd.so; touch speech_amd/libspeech_amd.so
  echo "== $v"; bash tools/ab_env.sh 1 - 2>&1
  python tools/bench_configs.py --only M-TIMIT 2>/dev/null | grep -A1 "AS SHIPPED\|ctc_config shapes" | grep "train_step_ms\|forward_ms" | tr '\n' ' '; echo
done; done | tee gpurun_out/r6ca/ab.txt
