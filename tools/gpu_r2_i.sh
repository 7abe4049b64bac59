#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2i; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1 )
f=$(find /tmp/prof_i -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(list(rows[0].keys()))
sel = [r for r in rows if "gemm_f32_kernel" in r["Kernel_Name"]]
for r in sel[-60:]:
    n = r["Kernel_Name"]; kind = n[n.index("<"):n.index(">") + 1]
    print(kind, "grid", r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), "wg", r.get("Workgroup_Size_X"), "lds", r.get("LDS_Block_Size"), "dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
