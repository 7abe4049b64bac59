#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|GRBM|TCC|TCP|TA)_[A-Z0-9_]+\b" | sort -u > $R/$O/counters.txt
wc -l $R/$O/counters.txt
pmc() { # tag counters... -- args
  local tag=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/gemm_one.py "$@" > $R/$O/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
agg = collections.defaultdict(list)
for r in rows:
    if "gemm_f32_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$tag", {k: round(sum(v) / len(v)) for k, v in agg.items()}, "launches", max(len(v) for v in agg.values()) if agg else 0)
PY
}
for shape in "4096 4096 4096 N T" "1536 512 15936 T N" "1024 512 1536 N N"; do
  tag=$(echo $shape | tr ' ' '_')
  pmc a_$tag SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $shape
  pmc b_$tag SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -- $shape
done
