"""Profiling target: Seq2Seq TRAIN steps only at BASELINE config 4 shapes (tools/bench_configs.m_seq2seq without its decode
legs), so that a kernel trace / stats of this script is the train step's and nothing else's.
    python tools/s2s_train_profile.py [iters] [dropout]"""
import json
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc  # noqa: E402
from speech_amd import ops  # noqa: E402
from speech_amd.models import Seq2Seq  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dropout = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
B, T, F, V, U = 16, 800, 161, 30, 100
cfg = {"dropout": dropout, "encoder": {"conv": [[32, 5, 8, 2], [32, 5, 8, 2]],
                                   "rnn": {"dim": 256, "layers": 4, "bidirectional": True}},
       "decoder": {"sample_prob": 0.2, "embedding_dim": 256, "log_t": True, "layers": 1}}
torch.manual_seed(2017)
random.seed(2017)
model = Seq2Seq(F, V + 2, cfg).cuda()
model.set_train()
flat_p, flat_g = model.flatten_parameters_()
rng = np.random.RandomState(2017)
inputs = tuple(rng.randn(T, F).astype(np.float32) for _ in range(B))
labels = tuple([V + 1] + list(rng.randint(0, V, U - 2)) + [V] for _ in range(B))
norm = torch.zeros(1, device="cuda")
out = {}


def step():
    model.zero_grad(set_to_none=True)
    loss = model.loss((inputs, labels))
    ops.backward(loss)
    ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0, norm_out=norm)
    out["loss"] = loss


sec = bc.timed(step, iters, warmup=4)
print(json.dumps({"train_step_ms": sec * 1e3, "loss": float(out["loss"].item())}))
