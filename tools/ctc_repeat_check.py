import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["SA_CTC_PROB"] = "3"; os.environ["SA_CTC_WIDE"] = "0"
import numpy as np, torch
from oracle import ctc_ref
from speech_amd import _lib
from speech_amd.ctc import ctc_loss_raw
rng = np.random.RandomState(41)
B, T, K = 4, 400, 11
ll = np.array([70, 130, 64, 100], dtype=np.int32)
labs = np.concatenate([np.full(70, 3), np.repeat(rng.randint(0, K - 1, 26), 5), np.repeat([1, 2], 32), rng.randint(0, 2, 100)]).astype(np.int32)
acts = rng.randn(B, T, K).astype(np.float32)
al = np.full(B, T, np.int32)
c, g = ctc_loss_raw(torch.from_numpy(acts).cuda(), torch.from_numpy(labs), torch.from_numpy(al), torch.from_numpy(ll))
torch.cuda.synchronize()
co, go = ctc_ref.ctc_loss(acts, labs, al, ll)
g = g.cpu().numpy(); c = c.cpu().numpy()
rows = np.abs(g.sum(axis=2))
print("cost", c, co)
for b in range(B):
    bad = np.nonzero(rows[b] > 1e-4)[0]
    print(b, "max defect %.2e" % rows[b].max(), "grad err %.2e" % np.abs(g[b] - go[b]).max(), "bad rows", (bad.min(), bad.max(), len(bad)) if bad.size else None)
al256 = lambda x: (x + 255) // 256 * 256
Lmax = int(ll.max()); nch = (Lmax + 1 + 63) // 64; Ppad = 64 * nch
o_stash = al256(B * ((T * K + 3) // 4 * 4) * 4)
n = _lib.lib().sa_ctc_workspace_bytes(T, Lmax, K, B)
ws = _lib.WORKSPACE.get(n, torch.device("cuda", 0), "ctc").view(torch.uint8)
raw = ws[o_stash:o_stash + B * T * 6 * Ppad * 4]
st = raw.view(torch.float32).cpu().numpy().reshape(B, T, 2, Ppad, 3)
ex = raw.view(torch.int32).cpu().numpy().reshape(B, T, 2, Ppad, 3)
for b, t in ((1, 385), (1, 394), (3, 386)):
    for d, name in ((0, "alpha"), (1, "beta")):
        v = st[b, t, d, :ll[b] + 1, :2]
        badp = np.nonzero(~np.isfinite(v).all(axis=1))[0]
        print("b", b, "t", t, name, "non-finite pairs", badp[:6].tolist(), "max hat %.3g" % np.nanmax(np.where(np.isfinite(v), v, 0)),
              "exps min/max", ex[b, t, d, :ll[b] + 1, 2].min(), ex[b, t, d, :ll[b] + 1, 2].max())
        for jp in badp[:2]:
            for tt in (t - 1, t, t + 1):
                print("     t", tt, "pair", jp, st[b, tt, d, jp - 1:jp + 2, :2].tolist(), ex[b, tt, d, jp - 1:jp + 2, 2].tolist())
b, t = 1, 385
x = acts[b].astype(np.float64); ly = x - x.max(1, keepdims=True); ly = (ly - np.log(np.exp(ly).sum(1, keepdims=True))) / np.log(2)
log2P = -co[b] / np.log(2)
Lb = int(ll[b]); lab_b = labs[70:70 + 130]
with np.errstate(divide="ignore"):
    aB, aL, aE = st[b, t, 0, :, 0].astype(np.float64), st[b, t, 0, :, 1].astype(np.float64), ex[b, t, 0, :, 2].astype(np.float64)
    bB, bL, bE = st[b, t, 1, :, 0].astype(np.float64), st[b, t, 1, :, 1].astype(np.float64), ex[b, t, 1, :, 2].astype(np.float64)
    xb = np.log2(aB[:Lb + 1]) + np.log2(bB[:Lb + 1]) + aE[:Lb + 1] + bE[:Lb + 1] - ly[t, K - 1] - log2P
    xl = np.log2(aL[:Lb]) + np.log2(bL[1:Lb + 1]) + aE[:Lb] + bE[1:Lb + 1] - ly[t, lab_b] - log2P
print("row", t, "max log2 gamma blank %.3f at pair %d; label %.3f at %d" % (np.nanmax(xb), np.nanargmax(xb), np.nanmax(xl), np.nanargmax(xl)))
j = int(np.nanargmax(xl))
print("  pair", j, "aL", aL[j], "aE", aE[j], "bL(pair j+1)", bL[j + 1], "bE", bE[j + 1], "bB", bB[j + 1], "sum gamma", np.nansum(2.0 ** xb) + np.nansum(2.0 ** xl))
for tt in range(399, 380, -1):
    print("t", tt, "beta pairs 124..130 hatB", [float("%.3g" % v) for v in st[1, tt, 1, 124:131, 0]], "hatL", [float("%.3g" % v) for v in st[1, tt, 1, 124:131, 1]], "e", ex[1, tt, 1, 124:131, 2].tolist())
