"""
oracle/torch_ref.py -- the reference's CTC train step restated with the SAME torch.nn CPU modules the reference
builds (so it times what the reference's CPU path actually executes), plus the C CTC restatement for the loss.
TEST INFRASTRUCTURE ONLY: used by tests/ (pinned to tests/golden/encoder_*.npz) and as bench.py's cpu_baseline
("port": /root/reference is not present on the GPU box and its warp-ctc dependency is not vendored at all).

Restates /root/reference/speech/models/model.py:12-42,60-79 (Conv2d+ReLU stack, nn.GRU batch_first, bi-sum),
ctc_model.py:17-19,25-40 (fc to |V|+1, CTC loss on raw logits, blank = last) and train.py:28-35
(zero_grad, loss, backward, clip_grad_norm(200), SGD step).  The CTC loss/gradient is oracle/ctc_ref.c (float
log-space, OpenMP over utterances -- the threading model of warp-ctc's CPU path), mean over the batch.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ctc_ref


class TorchRefCTC(nn.Module):
    def __init__(self, freq_dim, output_dim, config):
        super().__init__()
        convs, in_c = [], 1
        for out_c, h, w, s in config["encoder"]["conv"]:
            convs.extend([nn.Conv2d(in_c, out_c, (h, w), stride=(s, s), padding=0), nn.ReLU()])
            if config["dropout"] != 0:
                convs.append(nn.Dropout(p=config["dropout"]))
            in_c = out_c
        self.conv = nn.Sequential(*convs)
        f = freq_dim
        for out_c, h, w, s in config["encoder"]["conv"]:
            f = int(math.ceil((f - w + 1) / s))
        rnn = config["encoder"]["rnn"]
        self.rnn = nn.GRU(input_size=out_c * f, hidden_size=rnn["dim"], num_layers=rnn["layers"], batch_first=True,
                          dropout=config["dropout"], bidirectional=rnn["bidirectional"])
        self.fc = nn.Module()
        self.fc.fc = nn.Linear(rnn["dim"], output_dim + 1)
        self.blank = output_dim

    def encode(self, x):
        x = self.conv(x.unsqueeze(1))
        x = torch.transpose(x, 1, 2).contiguous()
        b, t, f, c = x.size()
        x, _ = self.rnn(x.view((b, t, f * c)))
        if self.rnn.bidirectional:
            half = x.size()[-1] // 2
            x = x[:, :, :half] + x[:, :, half:]
        return x

    def forward(self, x):
        return self.fc.fc(self.encode(x))


class _CTCRef(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, act_lens, label_lens, blank, threads):
        costs, grads = ctc_ref.ctc_loss(logits.detach().numpy(), labels, act_lens, label_lens, blank=blank,
                                        dtype=np.float32, num_threads=threads)
        B = logits.shape[0]
        ctx.g = torch.from_numpy(grads) / B
        return torch.tensor([costs.sum() / B], dtype=torch.float32)

    @staticmethod
    def backward(ctx, go):
        return ctx.g * go.reshape(()), None, None, None, None, None


def train_step(model, opt, x, labels, label_lens, threads=0):
    """One train.py:28-35 step on CPU.  x float32 (B,T,F); labels flat int32.  Returns (loss, grad_norm)."""
    opt.zero_grad()
    logits = model(x)
    B, Tp, _ = logits.shape
    loss = _CTCRef.apply(logits, labels, np.full(B, Tp, np.int32), label_lens, model.blank, threads)
    loss.backward()
    gn = nn.utils.clip_grad_norm_(model.parameters(), 200)
    opt.step()
    return float(loss.item()), float(gn)


# ---- Transducer (speech/models/transducer_model.py) -------------------------------------------------------------------
from . import transducer_ref  # noqa: E402


class TorchRefTransducer(TorchRefCTC):
    """The reference's Transducer restated with the same torch.nn CPU modules (transducer_model.py:14-78): encoder of
    model.py, nn.Embedding, nn.GRU prediction network over [0, y], fc1(x)[:, :, None] + fc1(y)[:, None], relu, fc2,
    log_softmax.  Pinned to the live reference by tests/golden/transducer_tiny.npz."""

    def __init__(self, freq_dim, vocab_size, config):
        super().__init__(freq_dim, vocab_size, config)
        del self.fc
        dec = config["decoder"]
        H = config["encoder"]["rnn"]["dim"]
        self.embedding = nn.Embedding(vocab_size, dec["embedding_dim"])
        self.dec_rnn = nn.GRU(input_size=dec["embedding_dim"], hidden_size=H, num_layers=dec["layers"],
                              batch_first=True, dropout=config["dropout"])
        self.blank = vocab_size
        self.fc1, self.fc2 = nn.Module(), nn.Module()
        self.fc1.fc = nn.Linear(H, H)
        self.fc2.fc = nn.Linear(H, vocab_size + 1)

    def forward(self, x, y_mat):
        x = self.encode(x)
        y = self.embedding(y_mat)
        b, t, h = y.shape
        y = torch.cat([torch.zeros((b, 1, h), dtype=y.dtype), y], dim=1)
        y, _ = self.dec_rnn(y)
        out = self.fc1.fc(x.unsqueeze(2)) + self.fc1.fc(y.unsqueeze(1))
        out = self.fc2.fc(torch.relu(out))
        return torch.log_softmax(out, dim=3)


class _TransducerRef(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, labels, act_lens, label_lens, blank):
        costs, grads = transducer_ref.transducer_loss(log_probs.detach().numpy(), labels, act_lens, label_lens,
                                                      blank=blank)
        ctx.g = torch.from_numpy(grads / log_probs.shape[0]).to(log_probs.dtype)
        return torch.tensor([costs.sum() / log_probs.shape[0]], dtype=log_probs.dtype)

    @staticmethod
    def backward(ctx, go):
        return ctx.g * go, None, None, None, None


def transducer_loss(model, x, y_mat, labels, act_lens, label_lens):
    """mean-over-batch Transducer loss of the restated model (the reduction speech_amd.transducer defaults to)."""
    return _TransducerRef.apply(model(x, y_mat), labels, act_lens, label_lens, model.blank)
