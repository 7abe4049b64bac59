"""Profiling target (tools/gpu_run.sh profpy:tools/infer_profile.py NAME BATCH): CTC.infer at the reference's evaluation batch
sizes (eval.py:20-22 defaults to 8, examples/timit/README.md:56-58 recommends 1) -- NAME = slibri | timit."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from speech_amd.models import CTC

name = sys.argv[1] if len(sys.argv) > 1 else "timit"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
timit = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2], [32, 5, 32, 1]], "rnn": {"dim": 256, "layers": 4, "bidirectional": True}}}
cfg, freq, vocab, frames = (bench.S_LIBRI, 80, 28, 1000) if name == "slibri" else (timit, 161, 48, 300)
torch.manual_seed(2017)
model = CTC(freq, vocab, cfg).cuda()
model.set_eval()
rng = np.random.RandomState(11)
batch = (tuple(rng.randn(frames, freq).astype(np.float32) for _ in range(bs)), tuple([0, 1] for _ in range(bs)))
for _ in range(3):
    model.infer(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    model.infer(batch)
torch.cuda.synchronize()
print("%s B=%d: %.3f ms per CTC.infer call" % (name, bs, (time.perf_counter() - t0) / 10 * 1e3))
