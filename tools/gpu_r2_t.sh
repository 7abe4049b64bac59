#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_t && SA_GRU_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o p -- python $GRAFT_REPO_ROOT/tools/bi_bench.py slibri_bi > /tmp/rp.log 2>&1 )
f=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in ("gemm_f32", "persist", "fused", "reduce"))]
# last backward pass: find the last 4 bwd persist kernels
idx = [i for i, r in enumerate(sel) if "bwd_persist" in r["Kernel_Name"]][-4]
t0 = int(sel[idx]["Start_Timestamp"])
for r in sel[idx - 3: idx + 40]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:30]
    print("%-32s q%-2s grid %7s %4s %3s start %9.1f dur %8.1f" % (n, r["Queue_Id"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
