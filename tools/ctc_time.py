"""M-CTC call time (B=32, T=1000, V+1=29, L=100): python tools/ctc_time.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_amd.ctc import CTCLabels, ctc_loss_raw
B, T, V, L = 32, 1000, 28, 100
rng = np.random.RandomState(2017)
acts = torch.from_numpy(rng.randn(B, T, V + 1).astype(np.float32)).cuda()
lab = CTCLabels(rng.randint(0, V, B * L).astype(np.int32), np.full(B, T, np.int32), np.full(B, L, np.int32), acts.device)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(5): ctc_loss_raw(acts, lab)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): ctc_loss_raw(acts, lab)
e1.record(); torch.cuda.synchronize()
print("SA_CTC_PROB=%s  %.1f us per call" % (os.environ.get("SA_CTC_PROB"), e0.elapsed_time(e1) / n * 1e3))
