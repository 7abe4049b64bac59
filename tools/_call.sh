# scratch: the command list of the current gpurun call (overwritten per call; see tools/README.md)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6bs
bash tools/ab_env.sh 3 SA_GRU_EXP=64 - 2>&1 | tee gpurun_out/r6bs/ab.txt
export TRACE_MIN_US=0
bash tools/gpu_run.sh r6bs "trace:tools/step_bench.py --no-prof" > /dev/null 2>&1
sed -n 1,3p gpurun_out/r6bs/step_bench_timeline.txt; grep -n "gru_bwd\|pk_pack\|side_wait" gpurun_out/r6bs/step_bench_timeline.txt | cut -c1-140
bash tools/gpu_run.sh r6bs "tests:baseline_configs or dropout or model or train_eval"
