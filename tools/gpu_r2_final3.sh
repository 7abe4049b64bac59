#!/bin/bash
# final tree: full GPU suite, smoke, headline bench (+ CPU baseline), rocprof kernel stats, all configs, batch sweep
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
bash tools/gpu_full.sh r02final 2>&1 | tail -22
O=gpurun_out/r02prof; mkdir -p $O
timeout 400 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err
timeout 200 python tools/uni_bench.py > $O/uni_bench.txt 2>&1; tail -3 $O/uni_bench.txt
python - <<PY
import json
c = json.load(open("$O/configs.json"))
for k in ("M-CTC", "M-STEP", "M-TIMIT", "M-RNNT", "M-S2S"):
    for r in c[k]:
        print(k, r.get("workload", r.get("L")), r.get("ms", r.get("train_step_ms")), r.get("forward_ms"))
PY
