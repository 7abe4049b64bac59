import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SA_CTC_PROB"] = "3"
import numpy as np, torch
from speech_amd import _lib
from speech_amd.ctc import ctc_loss_raw
rng = np.random.RandomState(1)
al256 = lambda x: (x + 255) // 256 * 256
T, B, K, L = 300, 2, 29, 50
acts = rng.randn(B, T, K).astype(np.float32)
labs = rng.randint(0, K - 1, B * L).astype(np.int32)
al, ll = np.full(B, T, np.int32), np.full(B, L, np.int32)
c, g = ctc_loss_raw(torch.from_numpy(acts).cuda(), torch.from_numpy(labs), torch.from_numpy(al), torch.from_numpy(ll))
torch.cuda.synchronize()
print("costs", c.cpu().numpy())
nch = 1
Ppad = 64 * nch
o_stash = al256(B * ((T * K + 3) // 4 * 4) * 4)
n = _lib.lib().sa_ctc_workspace_bytes(T, L, K, B)
ws = _lib.WORKSPACE.get(n, torch.device("cuda", 0), "ctc").view(torch.uint8)
raw = ws[o_stash:o_stash + B * T * 6 * Ppad * 4]
st = raw.view(torch.float32).cpu().numpy().reshape(B, T, 6, Ppad)
ex = raw.view(torch.int32).cpu().numpy().reshape(B, T, 6, Ppad)
b = 0
with np.errstate(divide="ignore"):
    for t in list(range(0, 12)) + list(range(12, T, 24)):
        la = np.log2(st[b, t, 0, :L + 1].astype(np.float64)) + ex[b, t, 4, :L + 1]
        ll2 = np.log2(st[b, t, 1, :L].astype(np.float64)) + ex[b, t, 4, :L]
        alive = np.nonzero(np.isfinite(la) | np.append(np.isfinite(ll2), False))[0]
        print("t", t, "alive pairs", (alive.min(), alive.max()) if alive.size else None, "max log2", np.nanmax(np.where(np.isfinite(la), la, -np.inf)),
              "e[0:6]", ex[b, t, 4, :6].tolist(), "hatB[0:4]", st[b, t, 0, :4].tolist())
