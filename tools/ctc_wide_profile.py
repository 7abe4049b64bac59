"""Profiling target: the saturating-batch M-CTC call (B=4096, T=1000, V+1=29, L=100), a few iterations."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd.ctc import CTCLabels, ctc_loss_raw  # noqa: E402

B, T, K, L = int(os.environ.get("B", 4096)), 1000, 29, 100
rng = np.random.RandomState(2017)
logits = torch.from_numpy(rng.randn(B, T, K).astype(np.float32)).cuda()
lab = CTCLabels(rng.randint(0, K - 1, B * L).astype(np.int32), np.full(B, T, np.int32), np.full(B, L, np.int32),
                logits.device)
for _ in range(int(os.environ.get("ITERS", 3))):
    ctc_loss_raw(logits, lab, blank=K - 1)
    ctc_loss_raw(logits, lab, blank=K - 1, want_grad=False)
torch.cuda.synchronize()
