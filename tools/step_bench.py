"""Train-step time of a CTC workload with the per-op profile (one tool for the uni / bidirectional / TIMIT shapes):
    python tools/step_bench.py [--case slibri|slibri_bi|timit|config2] [--dropout P] [--B N] [--steps K] [--no-prof]
Several --case values may be given.  Inputs resident in HBM, synthetic, seed 0."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from speech_amd import ops  # noqa: E402
from speech_amd.ctc import CTCLabels, CTCLoss  # noqa: E402
from speech_amd.models import CTC  # noqa: E402


def enc(conv, dim, layers, bi):
    return {"conv": conv, "rnn": {"dim": dim, "layers": layers, "bidirectional": bi}}


CASES = {  # F, V, B, T, L, encoder
    "slibri": (80, 28, 32, 1000, 100, enc([[32, 5, 32, 2]], 512, 4, False)),
    "slibri_bi": (80, 28, 32, 1000, 100, enc([[32, 5, 32, 2]], 512, 4, True)),
    "timit": (161, 48, 8, 300, 40, enc([[32, 5, 32, 2], [32, 5, 32, 1]], 256, 4, True)),
    "config2": (40, 61, 32, 1000, 100, enc([[32, 5, 32, 2]], 256, 2, False)),
    # eligibility: widths beyond the XCD-local kernels' register-resident weight fragments (H <= 512)
    "h768": (80, 28, 32, 1000, 100, enc([[32, 5, 32, 2]], 768, 4, False)),
    "h1024": (80, 28, 32, 1000, 100, enc([[32, 5, 32, 2]], 1024, 4, False)),
    "h640": (80, 28, 32, 1000, 100, enc([[32, 5, 32, 2]], 640, 4, False)),
}
ap = argparse.ArgumentParser()
ap.add_argument("--case", action="append")
ap.add_argument("--dropout", type=float, default=0.0)
ap.add_argument("--B", type=int, default=0)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--no-prof", action="store_true")
args = ap.parse_args()
for name in (args.case or ["slibri"]):
    F, V, B, T, L, e = CASES[name]
    B = args.B or B
    cfg = {"dropout": args.dropout, "encoder": e}
    torch.manual_seed(0)
    model = CTC(F, V, cfg).cuda()
    model.set_train()
    flat_p, flat_g = model.flatten_parameters_()
    rng = np.random.RandomState(0)
    x = torch.from_numpy(rng.randn(B, T, F).astype(np.float32)).cuda()
    Tp = model.conv_out_size(T, 0)
    lab = CTCLabels(rng.randint(0, V, B * L).astype(np.int32), np.full(B, Tp, np.int32), np.full(B, L, np.int32), x.device)
    loss_fn = CTCLoss(denom=B)

    def step():
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model.forward_impl(x), lab, None, None)
        ops.backward(loss)  # train.py:51 (loss.backward() without autograd's two scalar launches)
        ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0)
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        l = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    prof = {}
    if not args.no_prof:
        ops.PROFILE = ops.Profile()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        prof = {k: round(v["ms"] / 3, 3) for k, v in ops.PROFILE.summary().items()}
        ops.PROFILE = None
    print(name, json.dumps({"dropout": args.dropout, "B": B, "ms": round(dt * 1e3, 3), "utt_s": round(B / dt, 1),
                            "ms_per_utt": round(dt * 1e3 / B, 4), "loss": float(l.item()),
                            "status": ops.persist_status(), "prof": prof}))
