#!/bin/bash
# CTC: parity in all three modes + timing
O=gpurun_out/r2w; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ctc.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
cat > /tmp/w.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from speech_amd.ctc import CTCLabels, ctc_loss_raw
T, K, L = 1000, 29, 100
for B in (1024, 4096):
    rng = np.random.RandomState(2017)
    logits = torch.from_numpy(rng.randn(B, T, K).astype(np.float32)).cuda()
    lab = CTCLabels(rng.randint(0, K - 1, B * L).astype(np.int32), np.full(B, T, np.int32), np.full(B, L, np.int32), logits.device)
    for _ in range(2): ctc_loss_raw(logits, lab)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ctc_loss_raw(logits, lab)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print("SA_CTC_PROB=%s B=%d: %.3f ms  %.0f GB/s algorithmic" % (os.environ.get("SA_CTC_PROB"), B, ms, B * 232404 / ms / 1e6))
PY
for m in 1 0 3; do SA_CTC_PROB=$m timeout 300 python /tmp/w.py 2>&1 | grep -v amdgpu; done
