"""Two M-CTC calls (B=32 and B=4096; T=1000, V+1=29, L=100) for the PMC traffic passes: python tools/ctc_pmc_run.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_amd.ctc import CTCLabels, ctc_loss_raw
T, V, L = 1000, 28, 100
for B in (32, 4096):
    rng = np.random.RandomState(2017)
    acts = torch.from_numpy(rng.randn(B, T, V + 1).astype(np.float32)).cuda()
    lab = CTCLabels(rng.randint(0, V, B * L).astype(np.int32), np.full(B, T, np.int32), np.full(B, L, np.int32), acts.device)
    for _ in range(3):
        ctc_loss_raw(acts, lab)
    torch.cuda.synchronize()
    print("B", B, "algorithmic bytes per call", B * 232404)
