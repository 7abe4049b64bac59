#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py -x -q 2>&1 | tail -3 )
timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench.txt
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
run noov SA_GRU_OVERLAP=0 SA_GRU_POLL=0
run ov SA_GRU_POLL=0
run ov_e2 SA_GRU_POLL=0 SA_GRU_WG_EVERY=2
