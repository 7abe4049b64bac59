"""Wall time of forward / forward+backward of the bidirectional S-LIBRI encoder (python tools/bi_fwd_time.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_amd import ops
from speech_amd.ctc import CTCLabels, CTCLoss
from speech_amd.models import CTC
F, V, B, T, L = 80, 28, 32, 1000, 100
cfg = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 512, "layers": 4, "bidirectional": True}}}
torch.manual_seed(0)
model = CTC(F, V, cfg).cuda(); model.set_train()
flat_p, flat_g = model.flatten_parameters_()
rng = np.random.RandomState(0)
x = torch.from_numpy(rng.randn(B, T, F).astype(np.float32)).cuda()
Tp = model.conv_out_size(T, 0)
lab = CTCLabels(rng.randint(0, V, B * L).astype(np.int32), np.full(B, Tp, np.int32), np.full(B, L, np.int32), x.device)
loss_fn = CTCLoss(denom=B)
def fwd():
    with torch.no_grad():
        return model.forward_impl(x)
def step():
    model.zero_grad(set_to_none=True)
    loss = loss_fn(model.forward_impl(x), lab, None, None)
    loss.backward()
    ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0)
def wall(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("SA_GRU_FWD_CHUNKS=%s forward %.2f ms   train step %.2f ms" % (os.environ.get("SA_GRU_FWD_CHUNKS"), wall(fwd), wall(step)))
