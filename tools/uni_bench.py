"""Unidirectional S-LIBRI train step at a given batch size with the per-op profile: python tools/uni_bench.py [B ...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_amd import ops
from speech_amd.ctc import CTCLabels, CTCLoss
from speech_amd.models import CTC
F, V, T, L = 80, 28, 1000, 100
cfg = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 512, "layers": 4, "bidirectional": False}}}
for B in [int(a) for a in sys.argv[1:]] or [32, 64]:
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
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    ops.PROFILE = ops.Profile()
    for _ in range(3): step()
    torch.cuda.synchronize()
    prof = {k: round(v["ms"] / 3, 3) for k, v in ops.PROFILE.summary().items()}
    ops.PROFILE = None
    print("B=%d: %.2f ms/step = %.0f utt/s (%.3f ms per utterance)  %s" % (B, dt * 1e3, B / dt, dt * 1e3 / B, json.dumps(prof)))
