#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2j; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_health_dist.py -x -q 2>&1 | tail -4 )
python - <<'PY'
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from speech_amd import ops
from speech_amd.ctc import CTCLabels, CTCLoss
from speech_amd.models import CTC
cfg = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 512, "layers": 4, "bidirectional": False}}}
for B in (32, 48, 64, 96, 128):
    torch.manual_seed(0)
    model = CTC(80, 28, cfg).cuda(); model.set_train()
    flat_p, flat_g = model.flatten_parameters_()
    rng = np.random.RandomState(0)
    x = torch.from_numpy(rng.randn(B, 1000, 80).astype(np.float32)).cuda()
    lab = CTCLabels(rng.randint(0, 28, B * 100).astype(np.int32), np.full(B, 498, np.int32), np.full(B, 100, np.int32), x.device)
    loss_fn = CTCLoss(denom=B)
    def step():
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model.forward_impl(x), lab, None, None)
        loss.backward()
        ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print("B=%d  %.2f ms/step  %.0f utt/s  status %d" % (B, dt * 1e3, B / dt, ops.persist_status()))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['kernel_time_ms_per_step'])"
