import os, sys
os.environ["SA_CTC_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_amd import _lib
from speech_amd.ctc import ctc_loss_raw
for (B, L) in [(32, 100), (32, 60), (1, 60)]:
    T, K = 1000, 29
    rng = np.random.RandomState(0)
    acts = torch.from_numpy(rng.randn(B, T, K).astype(np.float32)).cuda()
    labs = torch.from_numpy(rng.randint(0, K - 1, B * L).astype(np.int32))
    al = torch.full((B,), T, dtype=torch.int32); ll = torch.full((B,), L, dtype=torch.int32)
    for want_grad in (False, True):
        for _ in range(3): ctc_loss_raw(acts, labs, al, ll, want_grad=want_grad)
        torch.cuda.synchronize()
        ws = _lib.WORKSPACE._bufs[(str(acts.device), "ctc")]
        lib = _lib.lib()
        # goffs region offset = sum of the aligned ly2, stash, lp regions (mirror of ctc_ws_layout)
        al256 = lambda x: (x + 255) // 256 * 256
        nch = (L + 1 + 63) // 64
        off = al256(B * ((T * K + 3) // 4 * 4) * 4) + al256(B * T * 6 * nch * 64 * 4) + al256(B * 8)
        v = ws[off:off + 24].cpu().numpy().view(np.uint64)
        cyc, ticks, steps = int(v[0]), int(v[1]), int(v[2])
        print("B=%d L=%d grad=%s: %.1f cycles/step, %.1f ns/step, effective clock %.2f GHz" % (B, L, want_grad, cyc / steps, ticks * 10.0 / steps, cyc / (ticks * 10.0)))
