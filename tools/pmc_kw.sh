cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4pmc
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_TRANS"; do
  first=${set%% *}
  rm -rf /tmp/pk_$first
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pk_$first -o p -- python $R/tools/ctc_b4096_time.py 4096 > $R/gpurun_out/r4pmc/log_$first.txt 2>&1
  python $R/tools/pmc_summary.py /tmp/pk_$first ctc_wave_p >> $R/gpurun_out/r4pmc/kw.txt 2>&1
done
cat $R/gpurun_out/r4pmc/kw.txt
