#!/bin/bash
# SQ / TA / TCC counters of the two fused recurrence kernels in the headline step (separate --pmc passes, --kernel-trace only)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/pmcgru; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
pass() {
  local tag=$1; shift
  rm -rf /tmp/pmc_$tag
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/pmc_$tag.log 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gru_bwd_fused_kernel" in n or "gru_fwd_fused_kernel" in n:
            agg[n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("pass $tag")
for k, d in agg.items():
    print("  %s  (dispatches %d)" % (k, max(len(v) for v in d.values())))
    for c in sorted(d):
        print("      %-40s %14.1f" % (c, sum(d[c]) / len(d[c])))
PY
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE > $R/$O/summary.txt
pass mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum >> $R/$O/summary.txt
cat $R/$O/summary.txt
