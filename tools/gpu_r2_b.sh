#!/bin/bash
# GPU session B: where did the forward regression come from, and what do the kernels cost beside side-stream GEMMs?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    r = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    k = r["kernel_time_ms_per_step"]
    print("%-22s %.2f ms  fwd %.2f bwd %.2f  frac %.4f" % ("$name", r["ms_per_step"], k["gru_fwd_stack"], k["gru_bwd_stack"], r["roofline"]["frac"]))
except Exception as e:
    print("$name failed", e, open("$O/bench_$name.err").read()[-800:])
PY
}
run base_noov SA_GRU_OVERLAP=0
run noov_prio0 SA_GRU_OVERLAP=0 SA_GRU_PRIO=0
run noov_prio0_lds84 SA_GRU_OVERLAP=0 SA_GRU_PRIO=0 SA_GRU_LDS_KB=84
run noov_lds84 SA_GRU_OVERLAP=0 SA_GRU_LDS_KB=84
run ov_prio0 SA_GRU_PRIO=0
run ov_prio1 SA_GRU_PRIO=1
prof() {
  local name=$1; shift
  ( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rocprof_$name.log 2>&1 )
  find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  find /tmp/prof_$name -name "*kernel_trace.csv" -exec cp {} $O/${name}_kernel_trace.csv \;
}
prof noov SA_GRU_OVERLAP=0
prof ov SA_GRU_OVERLAP=1
ls -la $O
head -12 $O/noov_kernel_stats.csv
head -12 $O/ov_kernel_stats.csv
# keep the traces small: one step's worth
for n in noov ov; do python - <<PY
import csv
rows = list(csv.DictReader(open("$O/${n}_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[-900:]
with open("$O/${n}_trace_tail.csv", "w") as f:
    w = csv.writer(f); w.writerow(["kernel", "queue", "start_us", "dur_us"])
    t0 = int(keep[0]["Start_Timestamp"])
    for r in keep:
        w.writerow([r["Kernel_Name"][:60], r.get("Queue_Id", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3])
PY
rm -f $O/${n}_kernel_trace.csv
done
