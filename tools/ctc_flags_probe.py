"""How many utterances of an M-CTC batch the probability-domain pass of K_W flags (each one is redone by the log-domain kernel):
python tools/ctc_flags_probe.py [B ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import _lib  # noqa: E402
from speech_amd.ctc import CTCLabels, ctc_loss_raw  # noqa: E402

T, V, L = 1000, 28, 100
dev = torch.device("cuda", 0)
for B in (int(a) for a in (sys.argv[1:] or ["4096"])):
    for scale in (1.0, 4.0, 12.0):
        rng = np.random.RandomState(2017)
        acts = scale * torch.randn(B, T, V + 1, device=dev, generator=torch.Generator(device=dev).manual_seed(2017))
        lab = CTCLabels(rng.randint(0, V, B * L).astype(np.int32), np.full(B, T, np.int32), np.full(B, L, np.int32), dev)
        ctc_loss_raw(acts, lab)
        torch.cuda.synchronize()
        off = _lib.lib().sa_ctc_flags_offset(T, L, V + 1, B)
        ws = _lib.WORKSPACE.get(off + 4 * B, dev, "ctc")
        fl = ws.view(torch.uint8)[off:off + 4 * B].view(torch.int32).cpu().numpy()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ctc_loss_raw(acts, lab)
        e1.record()
        torch.cuda.synchronize()
        print("B", B, "logit scale", scale, "flagged", int((fl != 0).sum()), "of", B, "ms", round(e0.elapsed_time(e1) / 3, 4), flush=True)
