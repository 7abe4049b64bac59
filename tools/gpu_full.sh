#!/bin/bash
# Full GPU check: -m gpu suite, smoke, headline bench (with CPU baseline), rocprof kernel stats of the bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-full}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<PY
import json
r = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step", "loss_step0", "loss_rel_err", "ctc_loss_step_ms", "persist_status")})
print(r["roofline"]); print(r["cpu_baseline"]); print(r["kernel_time_ms_per_step"])
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1 )
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -16 $O/kernel_stats.csv | cut -c1-160
