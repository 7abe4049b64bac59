"""The loss and every gradient of a Seq2Seq batch with the MFMA-layout attention kernels (s2s.kernels = 3: score network and context forward,
one-launch backward) against the round-4 backward on the same forward (2) and the round-4 kernels throughout (0):
    python tools/s2s_bwd_compare.py    -> per tensor: max |diff|, max |want|"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import _lib  # noqa: E402
from speech_amd.models import Seq2Seq  # noqa: E402

for (dim, F, T, B, U, conv) in ((16, 20, 90, 3, 7, [[4, 5, 9, 2]]), (256, 40, 400, 4, 12, [[8, 5, 8, 2]])):
    cfg = {"dropout": 0.0, "encoder": {"conv": conv, "rnn": {"dim": dim, "bidirectional": True, "layers": 1}},
           "decoder": {"embedding_dim": dim, "layers": 1, "log_t": True}}
    torch.manual_seed(11)
    m = Seq2Seq(F, 12, cfg).cuda()
    m.set_train()
    rng = np.random.RandomState(3)
    inputs = tuple(rng.randn(T, F).astype(np.float32) for _ in range(B))
    labels = tuple([11] + list(rng.randint(0, 10, U - 2)) + [10] for _ in range(B))
    got = {}
    for one in (3, 2, 0):
        _lib.set_option("s2s.kernels", one)
        m.zero_grad(set_to_none=True)
        loss = m.loss((inputs, labels))
        loss.backward()
        got[one] = {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters()}
        got[one]["(loss)"] = np.float64(loss.item())
    _lib.set_option("s2s.kernels", 3)
    print("dim", dim)
    for n in got[3]:
        a, w, w0 = got[3][n], got[2][n], got[0][n]
        print("  %-28s diff %.3e  (vs round-4 forward too: %.3e)  max %.3e  finite %s" %
              (n, np.abs(a - w).max(), np.abs(a - w0).max(), np.abs(w).max(), np.isfinite(a).all()))
