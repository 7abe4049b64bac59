#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
echo "== overlap on (XCD-filtered side GEMMs)"; timeout 300 python tools/bi_bench.py 2>&1 | grep -v amdgpu
echo "== overlap off"; SA_GRU_OVERLAP=0 timeout 300 python tools/bi_bench.py 2>&1 | grep -v amdgpu
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_l -o p -- python $GRAFT_REPO_ROOT/tools/bi_bench.py slibri_bi > /tmp/rp.log 2>&1 )
f=$(find /tmp/prof_l -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in ("gemm_f32", "persist", "fused"))]
t0 = int(sel[-70]["Start_Timestamp"])
for r in sel[-70:]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:34]
    print("%-36s q%-2s grid %5s %4s %3s start %9.1f dur %8.1f" % (n, r["Queue_Id"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
