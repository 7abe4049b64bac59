cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6cd
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r.get('loss_step0'), r.get('loss_rel_err'), r['roofline'].get('kernel_us'), r.get('persist_status'))" | tee gpurun_out/r6cd/first.txt
timeout 900 bash tools/ab_env.sh 3 SA_GRU_EXP=64 - 2>&1 | tee gpurun_out/r6cd/ab.txt
timeout 1200 bash tools/gpu_run.sh r6cd "tests:baseline_configs or dropout or model or train_eval or fused or wgrad or shared"
