import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from speech_amd import _lib
from speech_amd.ctc import CTCLabels, ctc_loss_raw
T, K, L = 1000, 29, 100
for B in (512, 1024, 4096):
    rng = np.random.RandomState(2017)
    logits = torch.from_numpy(rng.randn(B, T, K).astype(np.float32)).cuda()
    lab = CTCLabels(rng.randint(0, K - 1, B * L).astype(np.int32), np.full(B, T, np.int32), np.full(B, L, np.int32), logits.device)
    c, g = ctc_loss_raw(logits, lab)
    torch.cuda.synchronize()
    off = _lib.lib().sa_ctc_flags_offset(T, L, K, B)
    ws = _lib.WORKSPACE.get(off + 4 * B, logits.device, "ctc")
    fl = ws.view(torch.uint8)[off:off + 4 * B].view(torch.int32).cpu().numpy()
    rows = g.sum(dim=2).abs()
    print("B", B, "flags nonzero", int((fl != 0).sum()), "values", np.unique(fl).tolist(), "max row defect %.2e" % float(rows.max()), "cost[0:3]", c[:3].tolist())
    bad = np.nonzero(fl)[0]
    for bb in bad[:2]:
        r = rows[bb].cpu().numpy()
        idx = np.nonzero(r > 1e-3)[0]
        print("   utterance", bb, "bad rows", (idx.min(), idx.max(), len(idx)) if idx.size else None, "defects at t=0,100,..:", np.round(r[::100], 4).tolist())
        lab_b = lab.d_lab.cpu().numpy()[bb * L:(bb + 1) * L]
        print("   repeats in labels:", int((lab_b[1:] == lab_b[:-1]).sum()))
