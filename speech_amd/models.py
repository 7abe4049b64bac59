"""speech_amd.models -- the reference's model API on the HIP hot path.

Mirrors (names, arguments, return types, state-dict keys):
  /root/reference/speech/models/model.py      Model(input_dim, config), conv_out_size, encode, set_eval/set_train,
                                              is_cuda, encoder_dim; LinearND; zero_pad_concat
  /root/reference/speech/models/ctc_model.py  CTC(freq_dim, output_dim, config): forward, forward_impl, loss, collate,
                                              infer, max_decode, blank, fc

Parameters live in torch.nn containers (nn.Conv2d / nn.GRU / nn.Linear are used ONLY as parameter holders: they give
the reference's state-dict names -- conv.0.weight, rnn.weight_ih_l0, fc.fc.weight ... -- and, under the same
torch seed, the reference's initial values).  Their forward() is never called: all compute is the HIP library.
A CPU model raises on forward: there is no CPU path.

Deviations from the reference, all supersets:
  * `.volatile` is a dead torch-0.3 flag; set_eval()/infer() run under torch.no_grad() instead.
  * loss() returns a 1-element tensor, so both `loss.data[0]` (train.py:33) and `loss.item()` work.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ctc, decoder
from .encoder import EncoderFunction, EncoderPlan


class Model(nn.Module):

    def __init__(self, input_dim, config):
        super().__init__()
        self.input_dim = input_dim
        self._ctor_args = (input_dim, config)
        encoder_cfg = config["encoder"]
        convs = []
        in_c = 1
        for out_c, h, w, s in encoder_cfg["conv"]:
            convs.extend([nn.Conv2d(in_c, out_c, (h, w), stride=(s, s), padding=0), nn.ReLU()])
            if config["dropout"] != 0:
                convs.append(nn.Dropout(p=config["dropout"]))
            in_c = out_c
        self.conv = nn.Sequential(*convs)
        conv_out = out_c * self.conv_out_size(input_dim, 1)
        assert conv_out > 0, "Convolutional ouptut frequency dimension is negative."
        rnn_cfg = encoder_cfg["rnn"]
        self.rnn = nn.GRU(input_size=conv_out, hidden_size=rnn_cfg["dim"], num_layers=rnn_cfg["layers"],
                          batch_first=True, dropout=config["dropout"], bidirectional=rnn_cfg["bidirectional"])
        self._encoder_dim = rnn_cfg["dim"]
        self._plan = EncoderPlan(input_dim, config)
        self.volatile = False

    # ---- reference API -------------------------------------------------------------------------------------------
    def conv_out_size(self, n, dim):
        """model.py:44-52"""
        for c in self.conv.children():
            if type(c) == nn.Conv2d:
                k = c.kernel_size[dim]
                s = c.stride[dim]
                n = int(math.ceil((n - k + 1) / s))
        return n

    def forward(self, batch):
        raise NotImplementedError

    def loss(self, x, y):
        raise NotImplementedError

    def infer(self, x):
        raise NotImplementedError

    def set_eval(self):
        self.eval()
        self.volatile = True

    def set_train(self):
        self.train()
        self.volatile = False

    @property
    def is_cuda(self):
        return list(self.parameters())[0].is_cuda

    @property
    def encoder_dim(self):
        return self._encoder_dim

    # ---- HIP path --------------------------------------------------------------------------------------------------
    def _encoder_params(self):
        ps = []
        for c in self.conv.children():
            if type(c) == nn.Conv2d:
                ps += [c.weight, c.bias]
        for l in range(self.rnn.num_layers):
            for sfx in ([""] + (["_reverse"] if self.rnn.bidirectional else [])):
                ps += [getattr(self.rnn, "%s_l%d%s" % (n, l, sfx))
                       for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        return ps

    def _run(self, x, head_w, head_b):
        """head(encode(x)) through the single fused autograd Function."""
        _lib.require_cuda(x, "input batch (move the model to the GPU: model.cuda())")
        return EncoderFunction.apply(self._plan, self.training, x, *(self._encoder_params() + [head_w, head_b]))

    def encode(self, x):
        """model.py:60-79: (B, T, F) -> (B, T', H).  Runs the encoder with an identity head."""
        H = self._encoder_dim
        eye = torch.eye(H, dtype=torch.float32, device=x.device)
        zero = torch.zeros(H, dtype=torch.float32, device=x.device)
        return self._run(x, eye, zero)

    def flatten_parameters_(self):
        """Re-home every parameter into ONE flat fp32 buffer and give each a slot in ONE flat gradient buffer: the
        fused clip+SGD step and the single RCCL all-reduce both operate on these.  Call after .cuda().
        Per step: p.grad = None for all p (model.zero_grad(set_to_none=True)); backward then fills every slot.
        Returns (flat_params, flat_grads)."""
        ps = [p for p in self.parameters()]
        n = sum(p.numel() for p in ps)
        dev = ps[0].device
        flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in ps:
            k = p.numel()
            flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = flat_p[off:off + k].view(p.shape)
            p._grad_slot = flat_g[off:off + k].view(p.shape)  # EncoderFunction.backward writes gradients here
            p.grad = None
            off += k
        self._flat = (flat_p, flat_g)
        return self._flat


class LinearND(nn.Module):
    """model.py:115-133: nn.Linear over the last dimension of an N-D input (parameter holder + HIP GEMM)."""

    def __init__(self, *args):
        super().__init__()
        self.fc = nn.Linear(*args)

    def forward(self, x):
        from . import ops
        size = list(x.size())
        out = _LinearFunction.apply(x.contiguous().view(-1, size[-1]), self.fc.weight, self.fc.bias)
        size[-1] = out.size(-1)
        return out.view(size)


class _LinearFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        from . import ops
        _lib.require_cuda(x, "x")
        ctx.save_for_backward(x, w)
        return ops.gemm(x, w, trans_b=True, bias=b)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        return ops.gemm(dy, w), ops.gemm(dy, x, trans_a=True), ops.colsum(dy)


def zero_pad_concat(inputs):
    """model.py:135-141"""
    max_t = max(inp.shape[0] for inp in inputs)
    shape = (len(inputs), max_t, inputs[0].shape[1])
    input_mat = np.zeros(shape, dtype=np.float32)
    for e, inp in enumerate(inputs):
        input_mat[e, :inp.shape[0], :] = inp
    return input_mat


class CTC(Model):
    def __init__(self, freq_dim, output_dim, config):
        super().__init__(freq_dim, config)
        self._ctor_args = (freq_dim, output_dim, config)
        # include the blank token (ctc_model.py:17-19): blank is the LAST class
        self.blank = output_dim
        self.fc = LinearND(self.encoder_dim, output_dim + 1)
        self.ctc_denominator = None  # data-parallel training sets the GLOBAL batch size here (speech_amd.dist)

    def forward(self, batch):
        x, y, x_lens, y_lens = self.collate(*batch)
        return self.forward_impl(x)

    def forward_impl(self, x, softmax=False):
        if self.is_cuda:
            x = x.cuda(non_blocking=True)
        x = self._run(x, self.fc.fc.weight, self.fc.fc.bias)
        if softmax:
            return torch.nn.functional.softmax(x, dim=2)
        return x

    def loss(self, batch):
        x, y, x_lens, y_lens = self.collate(*batch)
        with torch.set_grad_enabled(not self.volatile):
            out = self.forward_impl(x)
            loss_fn = ctc.CTCLoss(denom=self.ctc_denominator)
            return loss_fn(out, y, x_lens, y_lens)

    def collate(self, inputs, labels):
        max_t = max(i.shape[0] for i in inputs)
        max_t = self.conv_out_size(max_t, 0)
        x_lens = torch.IntTensor([max_t] * len(inputs))
        x = torch.from_numpy(zero_pad_concat(inputs))
        y_lens = torch.IntTensor([len(l) for l in labels])
        y = torch.IntTensor([int(l) for label in labels for l in label])
        return [x, y, x_lens, y_lens]

    def infer(self, batch):
        """ctc_model.py:55-60: prefix beam search with beam_size=1 over the full padded T' of every utterance --
        on the device, from the logits (the softmax of :30-31 is fused into the decode kernel)."""
        x, y, x_lens, y_lens = self.collate(*batch)
        with torch.no_grad():
            logits = self.forward_impl(x)
            return decoder.beam_decode(logits, beam_size=1, blank=self.blank, input_is_logits=True)[0]

    @staticmethod
    def max_decode(pred, blank):
        """ctc_model.py:62-70 (host helper on a label path; the device version is decoder.greedy_decode)."""
        pred = list(pred)
        if not pred:
            return []
        prev = pred[0]
        seq = [prev] if prev != blank else []
        for p in pred[1:]:
            if p != blank and p != prev:
                seq.append(p)
            prev = p
        return seq
