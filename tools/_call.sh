# scratch: the command list of the current gpurun call (overwritten per call; see tools/README.md)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6bq
bash tools/ab_env.sh 3 - SA_GRU_EXP=64 SA_GRU_EXP=128 SA_GRU_EXP=256 2>&1 | tee gpurun_out/r6bq/ab.txt
for e in 128 256; do SA_GRU_EXP=$e python tools/gru_bwd_timing.py 2>&1 | grep "all blocks" | tee -a gpurun_out/r6bq/timing.txt; done
