cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s22
for sh in 0 1 2 3; do echo "threshold 8 GF >> $sh"; SA_GRU_EXP=$((sh * 1048576)) timeout 600 python tools/bench_configs.py --only M-TIMIT,M-S2S 2>/dev/null | grep -E "workload|train_step_ms" | paste - - | cut -c1-130; done | tee gpurun_out/r6s22/pk_threshold.txt
