#!/usr/bin/env python
"""tools/ctc_batch_sweep.py -- CTC loss fwd+bwd time vs batch size for the two alpha/beta kernels:
K_B (one workgroup per utterance, latency regime) and K_W (one wave per utterance, throughput regime).
M-CTC rows: T=1000, V+1=29, L=100, seed 2017.  Prints one JSON line per batch size."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd.ctc import CTCLabels, ctc_loss_raw  # noqa: E402

T, K, L = 1000, 29, 100


def run(B, wide, iters):
    os.environ["SA_CTC_WIDE"] = "1" if wide else "0"
    rng = np.random.RandomState(2017)
    logits = torch.from_numpy(rng.randn(B, T, K).astype(np.float32)).cuda()
    lab = CTCLabels(rng.randint(0, K - 1, B * L).astype(np.int32), np.full(B, T, np.int32), np.full(B, L, np.int32),
                    logits.device)
    for _ in range(2):
        ctc_loss_raw(logits, lab, blank=K - 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ctc_loss_raw(logits, lab, blank=K - 1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for B in [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]:
    it = max(3, min(50, 4096 // B))
    nb, wd = run(B, False, it), run(B, True, it)
    alg = 2 * B * T * K * 4 + B * L * 4 + B * 4
    print(json.dumps({"B": B, "K_B_ms": round(nb, 4), "K_W_ms": round(wd, 4),
                      "best_utt_per_s": round(B / min(nb, wd) * 1e3), "best_algorithmic_GBps": round(alg / min(nb, wd) / 1e6, 1)}))
