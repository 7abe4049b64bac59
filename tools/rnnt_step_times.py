"""Per-phase wall times (synchronised) of RNN-T train steps at BASELINE config 5 shapes."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import ops  # noqa: E402
from speech_amd.models import Transducer  # noqa: E402

B, T, F, V, L = 32, 1000, 80, 28, 100
cfg = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 512, "layers": 4, "bidirectional": False}},
       "decoder": {"embedding_dim": 256, "layers": 1}}
torch.manual_seed(2017)
model = Transducer(F, V, cfg).cuda()
model.set_train()
flat_p, flat_g = model.flatten_parameters_()
rng = np.random.RandomState(2017)
inputs = tuple(rng.randn(T, F).astype(np.float32) for _ in range(B))
labels = tuple(rng.randint(0, V, L) for _ in range(B))
norm = torch.zeros(1, device="cuda")


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for it in range(6):
    t0 = sync()
    model.zero_grad(set_to_none=True)
    loss = model.loss((inputs, labels))
    t1 = sync()
    loss.backward()
    t2 = sync()
    ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0, norm_out=norm)
    t3 = sync()
    print("step %d: loss() %.2f ms, backward %.2f ms, clip+sgd %.2f ms" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))

# pipelined (no synchronisation inside): GPU-side step durations from events, host-side issue times
N = 10
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
torch.cuda.synchronize()
ev[0].record()
h0 = time.perf_counter()
for it in range(N):
    model.zero_grad(set_to_none=True)
    loss = model.loss((inputs, labels))
    loss.backward()
    ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0, norm_out=norm)
    ev[it + 1].record()
    host.append((time.perf_counter() - h0) * 1e3)
    h0 = time.perf_counter()
torch.cuda.synchronize()
print("gpu  ms/step:", " ".join("%.1f" % ev[i].elapsed_time(ev[i + 1]) for i in range(N)))
print("host ms/step:", " ".join("%.1f" % h for h in host))
