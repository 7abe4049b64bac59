#!/bin/bash
# rocprof kernel stats of the headline bench -> gpurun_out/$1/kernel_stats.csv (+ per-step table)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-prof}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
STEPS=${2:-10}
( cd /tmp && rm -rf /tmp/prof_$TAG && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1 )
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
nsteps = $STEPS + 3 + 3 + 1   # timed + warm-up + profile pass + the step-0 forward
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (approx, / %d): %.2f ms" % (nsteps, tot / nsteps / 1e6))
for r in rows[:22]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print("%-72s calls %5s  avg %9.1f us  per-step %7.3f ms  %5.1f%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / nsteps / 1e6, float(r["Percentage"])))
PY
