"""Edge shapes of the RNN-Transducer train step: encoder / prediction-network widths that the fused joint and the persistent
recurrences do not take (H = 4 .. 640, embedding 4 .. 300), one utterance, ragged label lengths incl. one label, 1 - 2 prediction
layers, dropout -- loss and every gradient of the default kernel selection against the unfused joint + one-launch-per-step
recurrences (gru.persist = 0), then the greedy decode.      python tools/rnnt_shape_sweep.py [case index]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import _lib  # noqa: E402
from speech_amd.models import Transducer  # noqa: E402

# (H, enc layers, B, T, F, E, dec layers, max label length, dropout)
cases = [(4, 1, 2, 50, 20, 4, 1, 3, 0.0), (8, 1, 1, 40, 20, 8, 1, 1, 0.0), (36, 2, 3, 60, 20, 20, 2, 5, 0.3),
         (64, 1, 5, 45, 24, 64, 1, 7, 0.0), (128, 2, 17, 40, 24, 100, 1, 4, 0.2), (320, 1, 2, 38, 24, 300, 2, 6, 0.0),
         (640, 1, 2, 36, 24, 64, 1, 2, 0.0), (256, 1, 4, 70, 24, 128, 1, 20, 0.0)]
if len(sys.argv) > 1:
    cases = [cases[int(sys.argv[1])]]
bad = 0
for (H, L, B, T, F, E, DL, UL, p) in cases:
    print("H %3d  L %d  B %2d  T %3d  E %3d  pred layers %d  labels <= %2d  dropout %.1f ..." % (H, L, B, T, E, DL, UL, p), end=" ", flush=True)
    cfg = {"dropout": p, "encoder": {"conv": [[8, 5, 8, 2]], "rnn": {"dim": H, "layers": L, "bidirectional": False}},
           "decoder": {"embedding_dim": E, "layers": DL}}
    torch.manual_seed(H + E)
    m = Transducer(F, 12, cfg).cuda()
    m.set_train()
    rng = np.random.RandomState(T)
    inputs = tuple(rng.randn(T - 2 * (i % 3), F).astype(np.float32) for i in range(B))
    labels = tuple(list(rng.randint(0, 12, max(1, UL - i % 3))) for i in range(B))
    got, loss = {}, {}
    for persist in (-1, 0):
        _lib.set_option("gru.persist", persist)
        torch.manual_seed(7)
        m.zero_grad(set_to_none=True)
        lo = m.loss((inputs, labels))
        lo.backward()
        torch.cuda.synchronize()
        loss[persist] = float(lo.item())
        got[persist] = {n: q.grad.detach().cpu().numpy().copy() for n, q in m.named_parameters() if q.grad is not None}
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
    hyp = m.infer((inputs, labels))
    if len(hyp) != B:
        bad += 1
        print("\n  INFER", hyp)
    print("loss %.5f (rel diff %.1e), worst gradient diff %.1e of max, infer ok" % (loss[-1], dl, worst))
print("FAILED: %d" % bad if bad else "all shapes agree")
sys.exit(1 if bad else 0)
