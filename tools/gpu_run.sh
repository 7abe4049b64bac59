#!/bin/bash
# tools/gpu_run.sh TAG STAGE [STAGE ...] -- the ONE parameterised runner for gpurun calls (replaces the per-experiment
# one-off scripts of round 2).  Output goes to gpurun_out/TAG/.  Stages (each bounded by its own timeout):
#   tests[:EXPR]     pytest -m gpu (optionally -k EXPR), -x -q
#   smoke            __graft_entry__.smoke()
#   bench[:ARGS]     bench.py (ARGS e.g. "--steps 20 --warmup 5 --no-cpu-baseline") -> bench.json
#   prof[:ARGS]      rocprofv3 --kernel-trace --stats of bench.py ARGS -> kernel_stats.csv
#   configs[:ONLY]   tools/bench_configs.py [--only ONLY] -> configs.json
#   py:SCRIPT ARGS   python SCRIPT ARGS -> SCRIPT-basename.log         (tools/*.py experiments)
#   profpy:SCRIPT ARGS  rocprofv3 --kernel-trace --stats of python SCRIPT ARGS -> SCRIPT-basename_kernel_stats.csv
#   trace:SCRIPT ARGS  rocprofv3 --kernel-trace of python SCRIPT ARGS -> SCRIPT-basename_timeline.txt (the last step's kernels in time;
#                    TRACE_MIN_US=0 lists every kernel, default: those of >= 25 us)
#   pmc:COUNTERS:ARGS  rocprofv3 --pmc COUNTERS --kernel-trace of bench.py ARGS -> pmc_<first counter>.csv
# Example: gpurun --timeout 1500 -- 'bash tools/gpu_run.sh r3a tests:dropout bench "configs:M-STEP,M-TIMIT"'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
TAG=${1:-run}; shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for stage in "$@"; do
  name=${stage%%:*}; arg=""; [[ "$stage" == *:* ]] && arg=${stage#*:}
  echo "=== $stage"
  case $name in
    tests)
      if [ -n "$arg" ]; then ( time timeout 2400 python -m pytest tests -m gpu -x -q -k "$arg" ) > $O/pytest_${arg//[^a-zA-Z0-9]/_}.log 2>&1; tail -5 $O/pytest_${arg//[^a-zA-Z0-9]/_}.log
      else ( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log; fi ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)
      timeout 600 python bench.py $arg > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
      python - <<PY
import json
try:
    r = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print({k: r.get(k) for k in ("value", "ms_per_step", "loss_step0", "loss_rel_err", "ctc_loss_step_ms", "persist_status", "train_loop_utt_s")})
    print(r.get("roofline")); print(r.get("cpu_baseline")); print(r.get("kernel_time_ms_per_step"))
except Exception as e:
    print("bench output unreadable:", e)
PY
      ;;
    prof)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $ROOT/bench.py ${arg:---steps 10 --warmup 3 --no-cpu-baseline} > $ROOT/$O/rocprof.log 2>&1 )
      find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
      head -14 $O/kernel_stats.csv | cut -c1-170 ;;
    configs)
      if [ -n "$arg" ]; then timeout 1500 python tools/bench_configs.py --only "$arg" > $O/configs.json 2> $O/configs.err
      else timeout 1800 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err; fi
      tail -c 300 $O/configs.err; grep -E '"workload"|train_step_ms|"ms"|beam|forward_ms' $O/configs.json | cut -c1-120 ;;
    py)
      script=${arg%% *}; rest=""; [[ "$arg" == *" "* ]] && rest=${arg#* }
      log=$O/$(basename ${script%.py}).log
      timeout 1200 python $script $rest > $log 2>&1; tail -40 $log ;;
    profpy)
      script=${arg%% *}; rest=""; [[ "$arg" == *" "* ]] && rest=${arg#* }
      base=$(basename ${script%.py})
      ( cd /tmp && rm -rf /tmp/pp_${TAG}_$base && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_${TAG}_$base -o p -- python $ROOT/$script $rest > $ROOT/$O/${base}_rocprof.log 2>&1 )
      find /tmp/pp_${TAG}_$base -name "*kernel_stats.csv" -exec cp {} $O/${base}_kernel_stats.csv \;
      tail -3 $O/${base}_rocprof.log | cut -c1-400; head -16 $O/${base}_kernel_stats.csv | cut -c1-170 ;;
    pmc)
      ctrs=${arg%%:*}; bargs=${arg#*:}; first=${ctrs%% *}
      ( cd /tmp && timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$first -o p -- python $ROOT/bench.py ${bargs:---steps 4 --warmup 2 --no-cpu-baseline} > $ROOT/$O/pmc_$first.log 2>&1 )
      find /tmp/pmc_${TAG}_$first -name "*counter_collection.csv" -exec cp {} $O/pmc_$first.csv \;
      python tools/pmc_summary.py $O/pmc_$first.csv 2>/dev/null | head -30 ;;
    trace)   # trace:SCRIPT ARGS -> rocprofv3 --kernel-trace (timestamps) of python SCRIPT ARGS, then tools/trace_timeline.py
      script=${arg%% *}; rest=""; [[ "$arg" == *" "* ]] && rest=${arg#* }
      base=$(basename ${script%.py})
      ( cd /tmp && rm -rf /tmp/tr_${TAG}_$base && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_${TAG}_$base -o p -- python $ROOT/$script $rest > $ROOT/$O/${base}_trace.log 2>&1 )
      f=$(find /tmp/tr_${TAG}_$base -name "*kernel_trace.csv" | head -1)
      python tools/trace_timeline.py $f ${TRACE_MARKER:-clip_sgd_kernel} ${TRACE_MIN_US:-25} > $O/${base}_timeline.txt 2>&1; head -150 $O/${base}_timeline.txt ;;
    *) echo "unknown stage $name" ;;
  esac
done
