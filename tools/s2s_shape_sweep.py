"""Edge shapes of the Seq2Seq attention kernels: the MFMA-layout kernels (s2s.kernels = 3) against the round-4 backward on the
same forward (2: every gradient to 2e-5 of max) and against the round-4 kernels throughout (0: loss to 2e-6, gradients to 2e-3 of
max, the all-but-cancelling location-conv gradients to 5e-2), for widths that are not a multiple of 16, one-chunk and short
utterances, one utterance, one token, every odd tap count the ABI takes.      python tools/s2s_shape_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import _lib  # noqa: E402
from speech_amd.models import NNAttention, Seq2Seq  # noqa: E402

bad = 0
cases = [(4, 20, 40, 2, 4, 15), (8, 20, 30, 1, 2, 3), (20, 20, 64, 3, 5, 5), (36, 20, 150, 2, 6, 1), (64, 20, 21, 5, 3, 7),
         (128, 24, 300, 3, 8, 15), (256, 24, 25, 2, 4, 9), (200, 24, 90, 2, 5, 13), (16, 20, 19, 4, 9, 11),
         (64, 20, 620, 2, 4, 15), (32, 20, 60, 33, 3, 15)]   # T' > 256 (two passes over the alignment); 33 utterances
if len(sys.argv) > 1:
    cases = [cases[int(sys.argv[1])]]
for (dim, F, T, B, U, KS) in cases:
    cfg = {"dropout": 0.0, "encoder": {"conv": [[4, 5, 8, 2]], "rnn": {"dim": dim, "bidirectional": True, "layers": 1}},
           "decoder": {"embedding_dim": dim, "layers": 1, "log_t": True}}
    torch.manual_seed(dim + KS)
    m = Seq2Seq(F, 12, cfg)
    m.attend = NNAttention(dim, kernel_size=KS, log_t=True)
    m = m.cuda()
    m.set_train()
    rng = np.random.RandomState(T)
    inputs = tuple(rng.randn(T - 2 * (i % 5), F).astype(np.float32) for i in range(B))
    labels = tuple([11] + list(rng.randint(0, 10, U - 2)) + [10] for _ in range(B))
    got, loss = {}, {}
    for k in (0, 2, 3):
        _lib.set_option("s2s.kernels", k)
        m.zero_grad(set_to_none=True)
        lo = m.loss((inputs, labels))
        lo.backward()
        loss[k] = float(lo.item())
        torch.cuda.synchronize()
        got[k] = {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters()}
    _lib.set_option("s2s.kernels", 3)
    Tp = m.conv_out_size(T, 0)
    worst_t, worst_l = 0.0, 0.0
    for n in got[3]:
        zero = n.endswith("attend.nn.1.fc.bias")
        a, w, w0 = got[3][n], got[2][n], got[0][n]
        ok = np.isfinite(a).all()
        et = np.abs(a - w).max() / max(np.abs(w).max(), 1e-3)
        el = np.abs(a - w0).max() / max(np.abs(w0).max(), 1e-3)
        if not zero:
            worst_t, worst_l = max(worst_t, et), max(worst_l, el if "attend.conv" not in n else el / 25)
        if not ok or (not zero and (et > 2e-5 or el > (5e-2 if "attend.conv" in n else 2e-3))):
            bad += 1
            print("  MISMATCH", n, et, el, ok)
    dl = abs(loss[3] - loss[0]) / abs(loss[0])
    if loss[3] != loss[2] or dl > 2e-6:
        bad += 1
        print("  LOSS", loss)
    print("H %3d  T' %3d  B %d  tokens %d  taps %2d: loss %.6f (rel diff %.1e), worst gradient diff %.1e same forward, %.1e round-4 forward"
          % (dim, Tp, B, U - 1, KS, loss[3], dl, worst_t, worst_l))
print("FAILED: %d" % bad if bad else "all shapes agree")
sys.exit(1 if bad else 0)
