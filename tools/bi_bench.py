"""Train-step time of the bidirectional workloads (S-LIBRI bidirectional, shipped TIMIT shapes): python tools/bi_bench.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_amd import ops
from speech_amd.ctc import CTCLabels, CTCLoss
from speech_amd.models import CTC

CASES = {
    "slibri_bi": (80, 28, 32, 1000, 100, {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 512, "layers": 4, "bidirectional": True}}}),
    "timit": (161, 48, 8, 300, 40, {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2], [32, 5, 32, 1]], "rnn": {"dim": 256, "layers": 4, "bidirectional": True}}}),
}
out = {}
for name in (sys.argv[1:] or list(CASES)):
    F, V, B, T, L, cfg = CASES[name]
    torch.manual_seed(0)
    model = CTC(F, V, cfg).cuda(); model.set_train()
    flat_p, flat_g = model.flatten_parameters_()
    rng = np.random.RandomState(0)
    x = torch.from_numpy(rng.randn(B, T, F).astype(np.float32)).cuda()
    Tp = model.conv_out_size(T, 0)
    lab = CTCLabels(rng.randint(0, V, B * L).astype(np.int32), np.full(B, Tp, np.int32), np.full(B, L, np.int32), x.device)
    loss_fn = CTCLoss(denom=B)
    def step():
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model.forward_impl(x), lab, None, None)
        loss.backward()
        ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0)
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    ops.PROFILE = ops.Profile()
    for _ in range(3): step()
    torch.cuda.synchronize()
    prof = {k: round(v["ms"] / 3, 3) for k, v in ops.PROFILE.summary().items()}
    ops.PROFILE = None
    out[name] = {"ms": round(dt * 1e3, 3), "utt_s": round(B / dt, 1), "loss": float(l.item()), "status": ops.persist_status(), "prof": prof}
    print(name, json.dumps(out[name]))
