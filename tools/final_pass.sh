#!/bin/bash
# tools/final_pass.sh TAG -- the round's PMC / timeline / config measurements in ONE gpurun call (everything the
# profiles/rNN_* files beside the bench line and the kernel stats are made from).  Counter passes are separate rocprofv3 runs
# with --kernel-trace only (never combined with other trace domains).
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-final}; O=$R/gpurun_out/$TAG; mkdir -p $O
BARGS="--steps 2 --warmup 1 --no-cpu-baseline --headline-only"
pass() {  # pass NAME "COUNTERS"
  rm -rf /tmp/fp_$1
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/fp_$1 -o p -- python $R/bench.py $BARGS > $O/pmc_$1.log 2>&1
}
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
python $R/tools/pmc_traffic.py /tmp/fp_fetch /tmp/fp_write gru_ > $O/hbm_traffic_gru.json 2> $O/hbm_traffic_gru.err
pass sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
pass mem "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum"
python $R/tools/pmc_busy.py /tmp/fp_sq ${PROFILE_PREFIX:-r06}_pmc_gru_fused_kernels.txt > $O/pipe_busy.log 2>&1; cp $R/profiles/pipe_busy.json $O/pipe_busy.json
python $R/tools/pmc_clock.py /tmp/fp_sq gemm_pk gru_ conv_ > $O/effective_clock.txt 2>&1
{ echo "pass sq"; python $R/tools/pmc_summary.py /tmp/fp_sq gru_; echo "pass mem"; python $R/tools/pmc_summary.py /tmp/fp_mem gru_; } > $O/pmc_gru_fused_kernels.txt 2>&1
{ echo "pass sq"; python $R/tools/pmc_summary.py /tmp/fp_sq gemm_pk; echo "pass mem"; python $R/tools/pmc_summary.py /tmp/fp_mem gemm_pk; } > $O/pmc_gemm_pk.txt 2>&1
cd $R
export TRACE_MIN_US=0
bash tools/gpu_run.sh $TAG "trace:tools/step_bench.py --no-prof" > /dev/null 2>&1
mv $O/step_bench_timeline.txt $O/train_step_timeline.txt
bash tools/gpu_run.sh $TAG "trace:tools/step_bench.py --case slibri_bi --dropout 0.2 --no-prof" > /dev/null 2>&1
mv $O/step_bench_timeline.txt $O/bidirectional_timeline.txt
bash tools/gpu_run.sh $TAG "profpy:tools/step_bench.py --case slibri_bi --dropout 0.2 --no-prof" > /dev/null 2>&1
bash tools/pmc_ctc_traffic.sh $TAG > $O/ctc_traffic.log 2>&1
timeout 900 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err
tail -c 300 $O/configs.err
cat $O/hbm_traffic_gru.json | head -40
head -30 $O/pmc_gru_fused_kernels.txt
head -3 $O/train_step_timeline.txt $O/bidirectional_timeline.txt
grep -E '"workload"|train_step_ms|"ms"|forward_ms' $O/configs.json | cut -c1-140 | head -60
