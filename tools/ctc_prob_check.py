"""Probability-domain CTC pass alone (SA_CTC_PROB=3) against the fp64 oracle: flags, row-sum defects, gradient error."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SA_CTC_PROB"] = sys.argv[1] if len(sys.argv) > 1 else "3"
import numpy as np, torch
from oracle import ctc_ref
from speech_amd import _lib
from speech_amd.ctc import ctc_loss_raw

def make(seed, B, T, K, Lmin, Lmax, scale=1.0):
    rng = np.random.RandomState(seed)
    acts = (scale * rng.randn(B, T, K)).astype(np.float32)
    ll = rng.randint(Lmin, Lmax + 1, B).astype(np.int32)
    labs = np.concatenate([rng.randint(0, K - 1, l) for l in ll]).astype(np.int32)
    return acts, labs, np.full(B, T, np.int32), ll

for name, args in {"mctc": (2017, 32, 1000, 29, 100, 100), "small": (29, 6, 160, 29, 10, 50),
                   "L70_T200": (3, 4, 200, 29, 70, 70), "L50_T1000": (5, 4, 1000, 29, 50, 50),
                   "L64_T200": (7, 4, 200, 29, 64, 64), "L63_T200": (7, 4, 200, 29, 63, 63),
                   "L100_T300": (9, 4, 300, 29, 100, 100)}.items():
    acts, labs, al, ll = make(*args)
    B, T, K = acts.shape
    c, g = ctc_loss_raw(torch.from_numpy(acts).cuda(), torch.from_numpy(labs), torch.from_numpy(al), torch.from_numpy(ll))
    torch.cuda.synchronize()
    c, g = c.cpu().numpy(), g.cpu().numpy()
    co, go = ctc_ref.ctc_loss(acts, labs, al, ll)
    off = _lib.lib().sa_ctc_flags_offset(T, int(ll.max()), K, B)
    ws = _lib.WORKSPACE.get(off + 4 * B, torch.device("cuda", 0), "ctc")
    fl = ws.view(torch.uint8)[off:off + 4 * B].view(torch.int32).cpu().numpy()
    rows = np.abs(g.sum(axis=2))
    print(name, "flags", fl.tolist()[:8], "max row defect %.2e (median %.2e)" % (rows.max(), np.median(rows)),
          "cost rel err %.2e" % np.abs(c / co - 1).max(), "grad err %.2e" % np.abs(g - go).max())
    worst = np.unravel_index(rows.argmax(), rows.shape)
    print("   worst row (b, t) =", worst, "defects along t for that b:", np.round(rows[worst[0], ::100], 6).tolist())
