import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SA_CTC_PROB"] = "3"
import numpy as np, torch
from oracle import ctc_ref
from speech_amd.ctc import ctc_loss_raw
rng = np.random.RandomState(1)
for T in (300, 400, 500, 600, 700, 800, 1000):
    B, K, L = 2, 29, 50
    acts = rng.randn(B, T, K).astype(np.float32)
    labs = rng.randint(0, K - 1, B * L).astype(np.int32)
    al, ll = np.full(B, T, np.int32), np.full(B, L, np.int32)
    c, g = ctc_loss_raw(torch.from_numpy(acts).cuda(), torch.from_numpy(labs), torch.from_numpy(al), torch.from_numpy(ll))
    c, g = c.cpu().numpy(), g.cpu().numpy()
    co, go = ctc_ref.ctc_loss(acts, labs, al, ll)
    rows = np.abs(g.sum(axis=2))[0]
    bad = np.nonzero(rows > 1e-3)[0]
    print(T, "cost", c, co, "first/last bad row", (bad.min(), bad.max()) if bad.size else None)
