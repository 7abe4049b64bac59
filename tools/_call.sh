cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6cp
for r in 1 2 3; do for v in old new; do
  cp tools/_ab_libs/$v.so speech_amd/libspeech_amd.so; touch speech_amd/libspeech_amd.so
  echo "== $v"; SA_GRU_EXP=128 bash tools/ab_env.sh 1 - 2>&1
  python tools/step_bench.py --case slibri_bi --dropout 0.2 --no-prof --steps 6 2>/dev/null | tail -1 | cut -c1-60
done; done | tee gpurun_out/r6cp/ab.txt
