cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<PY
import sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
r = bench.ctc_legs(dev)
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
PY
timeout 300 python -m pytest tests/test_gpu_ctc.py -x -q 2>&1 | tail -2
