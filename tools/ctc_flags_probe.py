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
def aligned(B, labels, margin, rng):
    """Logits a TRAINED model would give: N(0, 1) noise plus `margin` on the class of one monotonic alignment per utterance
    (every label held for 1 - 3 frames at a random position, blanks elsewhere)."""
    a = torch.randn(B, T, V + 1, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    cls = np.full((B, T), V, np.int64)  # blank = V
    for b in range(B):
        starts = np.sort(rng.choice(T // 4, L, replace=False)) * 4
        for i, t0 in enumerate(starts):
            cls[b, t0:t0 + rng.randint(1, 4)] = labels[b * L + i]
    a.scatter_add_(2, torch.from_numpy(cls).to(dev).unsqueeze(2), torch.full((B, T, 1), float(margin), device=dev))
    return a


for B in (int(a) for a in (sys.argv[1:] or ["4096"])):
    for scale in (1.0, 4.0, 12.0, -5.0, -10.0, -20.0):   # negative: aligned logits with that margin
        rng = np.random.RandomState(2017)
        labels = rng.randint(0, V, B * L).astype(np.int32)
        if scale > 0:
            acts = scale * torch.randn(B, T, V + 1, device=dev, generator=torch.Generator(device=dev).manual_seed(2017))
        else:
            acts = aligned(B, labels, -scale, rng)
        lab = CTCLabels(labels, np.full(B, T, np.int32), np.full(B, L, np.int32), dev)
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
        print("B", B, ("random logits x %g" % scale) if scale > 0 else ("aligned logits, margin %g" % -scale), "flagged", int((fl != 0).sum()), "(alpha-dead %d, beta %d)" % (int((fl == 1).sum()), int((fl == 2).sum())), "of", B, "ms", round(e0.elapsed_time(e1) / 3, 4), flush=True)
