#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
./tools/ubench/cumask_probe 2>&1 | tee $O/cumask.txt
timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "dW|dmid|dx layer0|i2h" | tee $O/gemm_bench.txt
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    r = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    k = r["kernel_time_ms_per_step"]
    print("%-26s %.2f ms  fwd %.2f bwd %.2f gemm %.2f conv %.2f+%.2f frac %.4f us/launch %.1f st %s" % ("$name", r["ms_per_step"], k["gru_fwd_stack"], k["gru_bwd_stack"], k["gemm"], k["conv_fwd"], k["conv_bwd"], r["roofline"]["frac"], r["roofline"]["avg_launch_us"], r["persist_status"]))
except Exception as e:
    print("$name failed", e, open("$O/bench_$name.err").read()[-800:])
PY
}
run default
timeout 600 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err; python - <<PY
import json
try:
    r = json.load(open("$O/configs.json"))
    for k, v in r.items():
        print(k, json.dumps(v)[:400])
except Exception as e:
    print("configs failed", e, open("$O/configs.err").read()[-1500:])
PY
