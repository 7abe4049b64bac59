#!/bin/bash
# GPU session A (round 2): full -m gpu suite, headline bench with / without the side-stream weight gradients, rocprof.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
O=gpurun_out/r2a
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_overlap.json 2> $O/bench_overlap.err
SA_GRU_OVERLAP=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_nooverlap.json 2> $O/bench_nooverlap.err
SA_GRU_WG_EVERY=2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_every2.json 2> $O/bench_every2.err
SA_GRU_WG_EVERY=8 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_every8.json 2> $O/bench_every8.err
for f in overlap nooverlap every2 every8; do python - <<PY
import json
try:
    r = json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", r["ms_per_step"], r["value"], r.get("loss_rel_err"), r.get("persist_status"), r["roofline"]["frac"] if r.get("roofline") else None)
    print("   ", {k: round(v, 3) for k, v in r["kernel_time_ms_per_step"].items()})
except Exception as e:
    print("$f failed", e, open("$O/bench_$f.err").read()[-1500:])
PY
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r2a -o r2a -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1
cd "$GRAFT_REPO_ROOT"
find /tmp/prof_r2a -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find /tmp/prof_r2a -name "*kernel_trace.csv" -exec sh -c 'head -4000 "$1" > '$O'/kernel_trace_head.csv' _ {} \;
head -30 $O/kernel_stats.csv
