"""Edge shapes of the CTC train step (the north-star path): widths the persistent recurrences do not take (4 .. 1024, odd multiples),
one utterance / ragged batches, short utterances, uni- and bidirectional stacks, with and without dropout -- loss and every
gradient of the default kernel selection against the one-launch-per-time-step kernels (gru.persist = 0), each case in THIS
process (a host crash shows as the last line printed).      python tools/ctc_shape_sweep.py [case index]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import _lib  # noqa: E402
from speech_amd.models import CTC  # noqa: E402

# (H, layers, bidirectional, B, T, F, conv, dropout)
cases = [(4, 1, True, 2, 60, 20, [[4, 5, 8, 2]], 0.0), (8, 2, False, 1, 40, 20, [[4, 5, 8, 2]], 0.0),
         (12, 2, True, 3, 70, 20, [[4, 5, 8, 2]], 0.3), (100, 1, True, 5, 50, 24, [[8, 5, 8, 2]], 0.0),
         (192, 2, False, 17, 45, 24, [[8, 5, 8, 2]], 0.2), (520, 1, True, 2, 40, 24, [[8, 5, 8, 2]], 0.0),
         (640, 2, False, 3, 38, 24, [[8, 5, 8, 2]], 0.0), (1024, 1, True, 1, 36, 24, [[8, 5, 8, 2]], 0.0),
         (128, 3, True, 33, 64, 40, [[8, 5, 8, 2], [8, 5, 8, 1]], 0.4), (256, 2, False, 1, 33, 40, [[32, 5, 32, 2]], 0.0),
         (64, 1, False, 2, 200, 40, [[32, 5, 32, 2]], 0.0), (384, 2, True, 9, 50, 24, [[8, 5, 8, 2]], 0.0),
         # one and two output frames (T' = 1, 2): a recurrence of a single step, labels of at most one symbol
         (256, 2, False, 3, 5, 24, [[8, 5, 8, 2]], 0.0), (128, 1, True, 2, 7, 24, [[8, 5, 8, 2]], 0.0),
         (512, 4, False, 32, 9, 24, [[8, 5, 8, 2]], 0.0),
         # more batch tiles than one pass of the persistent kernels hosts (B = 64, 100, 130)
         (512, 2, False, 64, 44, 24, [[8, 5, 8, 2]], 0.2), (256, 2, True, 100, 40, 24, [[8, 5, 8, 2]], 0.0),
         (128, 1, False, 130, 36, 24, [[8, 5, 8, 2]], 0.0)]
if len(sys.argv) > 1:
    cases = [cases[int(sys.argv[1])]]
bad = 0
for (H, L, bi, B, T, F, conv, p) in cases:
    print("H %4d  L %d  %s  B %2d  T %3d  conv %s  dropout %.1f ..." % (H, L, "bi " if bi else "uni", B, T, conv, p), end=" ", flush=True)
    cfg = {"dropout": p, "encoder": {"conv": conv, "rnn": {"dim": H, "layers": L, "bidirectional": bi}}}
    torch.manual_seed(H + L)
    m = CTC(F, 20, cfg).cuda()
    m.set_train()
    rng = np.random.RandomState(T)
    inputs = tuple(rng.randn(T - (3 * (i % 4) if T > 20 else 0), F).astype(np.float32) for i in range(B))
    labels = tuple(list(rng.randint(0, 19, (2 + i % 3) if T > 20 else 1)) for i in range(B))
    got, loss = {}, {}
    for persist in (-1, 0):
        _lib.set_option("gru.persist", persist)
        torch.manual_seed(7)   # the dropout masks' seed
        m.zero_grad(set_to_none=True)
        lo = m.loss((inputs, labels))
        lo.backward()
        torch.cuda.synchronize()
        loss[persist] = float(lo.item())
        got[persist] = {n: q.grad.detach().cpu().numpy().copy() for n, q in m.named_parameters()}
    _lib.set_option("gru.persist", -1)
    worst = 0.0
    for n in got[-1]:
        a, w = got[-1][n], got[0][n]
        e = np.abs(a - w).max() / max(np.abs(w).max(), 1e-3)
        worst = max(worst, e)
        if not np.isfinite(a).all() or e > 1e-3:
            bad += 1
            print("\n  MISMATCH", n, e)
    dl = abs(loss[-1] - loss[0]) / max(abs(loss[0]), 1e-6)
    if not np.isfinite(loss[-1]) or dl > 1e-5:
        bad += 1
        print("\n  LOSS", loss)
    m.set_eval()
    hyp = m.infer((inputs, labels))   # the beam-1 decode on the same shapes (a crash or a wrong count shows here)
    if len(hyp) != B or not all(all(0 <= int(t) < 20 for t in h) for h in hyp):
        bad += 1
        print("\n  INFER", hyp)
    print("loss %.5f (rel diff %.1e), worst gradient diff %.1e of max, infer ok" % (loss[-1], dl, worst))
print("FAILED: %d" % bad if bad else "all shapes agree")
sys.exit(1 if bad else 0)
