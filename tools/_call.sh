cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6ci
for r in 1 2; do for e in 0 256 512 768; do
  echo "== exp $e"; SA_GRU_EXP=$e python tools/step_bench.py --case slibri_bi --dropout 0.2 --no-prof --steps 6 2>/dev/null | tail -3
done; done | tee gpurun_out/r6ci/ab.txt
