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

    def encode(self, x, masks=None):
        """masks = None: the modules as the reference builds them (their own dropout in training mode).
        masks = {"conv": [m_i (B, O, T', F')], "gru": [m_l (B, T', D*H), l < layers - 1]}: the SAME computation with the
        Bernoulli factors given instead of drawn -- nn.Dropout(x) = x * m behind each ReLU (model.py:25-27), and
        nn.GRU(dropout=p) = layer l+1 reads (layer l output) * m_l (model.py:38) -- so that a HIP pass whose masks are
        known (oracle/philox_ref.py) can be checked value by value."""
        if masks is None:
            x = self.conv(x.unsqueeze(1))
        else:
            x, i = x.unsqueeze(1), 0
            for m in self.conv.children():
                if isinstance(m, nn.Dropout):
                    continue
                x = m(x)
                if isinstance(m, nn.ReLU):
                    x = x * torch.as_tensor(masks["conv"][i])
                    i += 1
        x = torch.transpose(x, 1, 2).contiguous()
        b, t, f, c = x.size()
        x = x.view((b, t, f * c))
        if masks is None:
            x, _ = self.rnn(x)
        else:
            r = self.rnn
            D = 2 if r.bidirectional else 1
            for l in range(r.num_layers):
                flat = [getattr(r, "%s_l%d%s" % (n, l, sfx)) for sfx in (("", "_reverse") if D == 2 else ("",))
                        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
                h0 = torch.zeros(D, b, r.hidden_size, dtype=x.dtype)
                x, _ = torch._VF.gru(x, h0, flat, True, 1, 0.0, False, r.bidirectional, True)
                if l + 1 < r.num_layers:
                    x = x * torch.as_tensor(masks["gru"][l])
        if self.rnn.bidirectional:
            half = x.size()[-1] // 2
            x = x[:, :, :half] + x[:, :, half:]
        return x

    def forward(self, x, masks=None):
        return self.fc.fc(self.encode(x, masks))


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

    def forward(self, x, y_mat, masks=None):
        """masks: the ENCODER's dropout factors (see TorchRefCTC.encode); a one-layer prediction network has no
        dropout site (nn.GRU drops between layers only, transducer_model.py:23-26)."""
        x = self.encode(x, masks)
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


def transducer_loss(model, x, y_mat, labels, act_lens, label_lens, masks=None):
    """mean-over-batch Transducer loss of the restated model (the reduction speech_amd.transducer defaults to)."""
    return _TransducerRef.apply(model(x, y_mat, masks), labels, act_lens, label_lens, model.blank)


# ---- Seq2Seq (speech/models/seq2seq.py) -------------------------------------------------------------------------------
class _RefNNAttention(nn.Module):
    """seq2seq.py:331-360 (NNAttention)."""

    def __init__(self, n_channels, kernel_size=15, log_t=False):
        super().__init__()
        self.conv = nn.Conv1d(1, n_channels, kernel_size, padding=(kernel_size - 1) // 2)
        lin = nn.Module()
        lin.fc = nn.Linear(n_channels, 1)
        self.nn = nn.Sequential(nn.ReLU(), lin)
        self.log_t = log_t

    def forward(self, eh, dhx, ax=None):
        pax = eh + dhx
        if ax is not None:
            pax = pax + self.conv(ax.unsqueeze(1)).transpose(1, 2)
        pax = self.nn[1].fc(torch.relu(pax)).squeeze(2)
        if self.log_t:
            pax = math.log(pax.size()[1]) * pax
        ax = torch.softmax(pax, dim=1)
        sx = torch.sum(eh * ax.unsqueeze(2), dim=1, keepdim=True)
        return sx, ax


class TorchRefSeq2Seq(TorchRefCTC):
    """The reference's Seq2Seq restated with the same torch.nn CPU modules (seq2seq.py:14-127): encoder of model.py,
    nn.Embedding, nn.GRUCell, NNAttention, fc to vocab_size - 1 classes, teacher forcing (scheduled sampling off).
    Pinned to the live reference by tests/golden/seq2seq_tiny.npz (logits, loss, every parameter gradient)."""

    def __init__(self, freq_dim, vocab_size, config):
        super().__init__(freq_dim, vocab_size, config)
        dec = config["decoder"]
        H = config["encoder"]["rnn"]["dim"]
        self.embedding = nn.Embedding(vocab_size, dec["embedding_dim"])
        self.dec_rnn = nn.GRUCell(input_size=dec["embedding_dim"], hidden_size=H)
        self.attend = _RefNNAttention(H, log_t=dec.get("log_t", False))
        self.fc = nn.Module()
        self.fc.fc = nn.Linear(H, vocab_size - 1)

    def decode_step(self, x, y, state=None):
        if state is None:
            hx, ax, sx = torch.zeros((x.shape[0], x.shape[2]), dtype=x.dtype), None, None
        else:
            hx, ax, sx = state
        ix = self.embedding(y)
        if sx is not None:
            ix = ix + sx
        hx = self.dec_rnn(ix.squeeze(1), hx)
        ox = hx.unsqueeze(1)
        sx, ax = self.attend(x, ox, ax)
        return self.fc.fc((ox + sx).squeeze(1)), (hx, ax, sx)

    def forward(self, x, y, masks=None, sample=None):
        """x (B, T, F) float, y (B, U) int64 with start / end tokens -> (logits (B, U-1, V-1), aligns (B, U-1, T')).
        masks: the encoder's dropout factors (TorchRefCTC.encode).  sample: per-token flags (index t >= 1) -- where set,
        token t's input is the argmax of token t-1's logits instead of y[:, t] (scheduled sampling, seq2seq.py:91-96;
        the reference draws random.random() < sample_prob once per token after the first)."""
        x = self.encode(x, masks)
        out, aligns, state = [], [], None
        for t in range(y.size()[1] - 1):
            inp = y[:, t:t + 1]
            if t > 0 and sample is not None and sample[t]:
                inp = torch.max(out[-1], dim=1)[1].unsqueeze(1)
            o, state = self.decode_step(x, inp, state)
            out.append(o)
            aligns.append(state[1])
        return torch.stack(out, dim=1), torch.stack(aligns, dim=1)

    def loss(self, x, y, masks=None, sample=None):
        out, _ = self(x, y, masks, sample)
        b, _, k = out.size()
        return nn.functional.cross_entropy(out.reshape(-1, k), y[:, 1:].reshape(-1), reduction="sum") / b

    def infer(self, x, y0, end_tok, max_len):
        """seq2seq.py:150-185: greedy decode from the start tokens y0 (B, 1)."""
        x = self.encode(x)
        y, state, seq = y0, None, [y0]
        for _ in range(max_len):
            o, state = self.decode_step(x, y, state)
            y = torch.max(o, dim=1)[1].unsqueeze(1)
            seq.append(y)
            if int(torch.sum(y == end_tok)) == y.numel():
                break
        return torch.cat(seq, dim=1)
